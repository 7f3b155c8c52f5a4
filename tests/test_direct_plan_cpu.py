"""Shape and budget arithmetic of the Direct N^2 path (nbody_hip_direct_info with a NULL context: plain host code
in n-body_amd/csrc/direct_sym.hip, no device needed) -- which kernel a call takes, with how many bodies per lane,
and what the deterministic form's slot planes cost (12 bytes per reaction slot, 24 per I-side slot, per body)."""
import ctypes as C

import pytest


def info(nb, n, eps2=1e-6):
    from nbody_amd._lib import DirectInfoStruct, check, load
    s = DirectInfoStruct()
    check(load().nbody_hip_direct_info(None, n, eps2, C.byref(s)))
    return s


def test_kernel_choice_by_size(nb):
    assert info(nb, 4096).kernel == 0 and info(nb, 12287).kernel == 0       # one-sided kernel below 12,288 bodies
    assert info(nb, 12288).kernel == 2                                       # symmetric kernel, slot planes
    assert info(nb, 1 << 20, eps2=0.0).kernel == 0                           # tiny softening: guarded one-sided kernel
    assert [info(nb, n).bodies_per_lane_equal for n in (12288, 27999, 28000, 99999, 100000, 199999, 200000, 1 << 20)] \
        == [6, 6, 8, 8, 12, 12, 16, 16]
    # general masses run with the same bodies per lane as equal masses (round 2: 12 where those take 16 -- the slot
    # kernel spilled; the single loop body of round 3 fits 246 registers)
    assert info(nb, 1 << 20).bodies_per_lane_general == 16 and info(nb, 150000).bodies_per_lane_general == 12


@pytest.mark.parametrize("n", [12288, 65536, 131072, 262144, 1 << 20, 1 << 21])
def test_slot_plane_bytes(nb, n):
    s = info(nb, n)
    assert s.kernel == 2 and s.deterministic_mode == 1 and s.last_kernel == -1 and s.workspace_bytes_held == 0

    def need(R):
        S = 256 * R
        NB = -(-n // S)
        D, plane = NB // 2, NB * S
        total = (D + 1) * (S // 64)
        splits = max(1, -(-256 * 16 // NB))
        per = max(4, -(-total // splits))
        if per > S // 64:
            per = -(-per // (S // 64)) * (S // 64)
        splits = -(-total // per)
        react = (D * 3 * plane * 4 + 255) // 256 * 256
        return D, splits, react + splits * 3 * plane * 8
    D, splits, b_eq = need(s.bodies_per_lane_equal)
    assert (s.reaction_slots, s.iside_slots) == (D, splits)
    assert s.workspace_bytes_needed == max(b_eq, need(s.bodies_per_lane_general)[2]) == s.slot_bytes_wanted


def test_headline_size_costs_under_2_gb_and_budget_switches_to_atomics(nb):
    s = info(nb, 1 << 20)
    # 128 reaction slots x 12 B + 15 I-side slots x 24 B per body = 1.9 GB for both instantiations
    # (round 2: 24 B per slot and a 12-bodies-per-lane general-mass shape with 183 slots: 4.6 GB reserved)
    assert s.reaction_slots == 128 and 1.85e9 < (128 * 12 + s.iside_slots * 24) * (1 << 20) < 2.0e9
    assert 1.85e9 < s.workspace_bytes_needed < 2.0e9
    big = info(nb, 1 << 22)
    assert big.kernel == 1 and big.slot_bytes_wanted > (24 << 30)            # beyond the 24 GiB budget: atomics ...
    assert big.workspace_bytes_needed == 3 * 8 * 4194304                     # ... 24 bytes per body
    with pytest.raises(Exception):
        info(nb, (1 << 30) + 1)
