// ref_observability_driver.cpp -- test infrastructure (oracle/): prints what the REFERENCE's own
// serializeBenchmarkRunRecords / PhaseProfiler (compiled from
// /root/reference/src/utils/performance_observability.cpp by Makefile.ref; host only) produce for a fixed
// set of records with awkward names and numbers, so that n-body_amd/observability.py can be
// compared with it byte for byte (tests/test_cli_observability_cpu.py builds the same records).
#include <cmath>
#include <cstdio>
#include <limits>
#include <string>
#include <vector>

#include "nbody/performance_observability.hpp"

using namespace nbody;

int main() {
  std::vector<BenchmarkRunRecord> records;
  BenchmarkRunRecord a;
  a.benchmark_name = "force.direct_n2";
  a.force_method = ForceMethod::DIRECT_N2;
  a.particle_count = 1048576;
  a.iterations = 20;
  a.metrics = {{"wall_time_ms", 156.793342}, {"steps_per_s", 6.3778218}, {"pair_interactions_per_s", 7.0124892e12},
               {"zero", 0.0}, {"neg_zero", -0.0}, {"tiny", 1.25e-7}, {"exact", 256.0}, {"third", 1.0 / 3.0}};
  a.parameters = {{"dt", 0.001}, {"softening", 0.1}, {"gpus", 1.0}, {"big", 1e21}, {"negative", -42.5}};
  PhaseProfiler prof;
  prof.record("simulation.update", Milliseconds(1.5));
  prof.record("force", Milliseconds(0.25));
  prof.record("simulation.update", Milliseconds(2.75));
  a.phase_timings = prof.snapshot();
  records.push_back(a);
  BenchmarkRunRecord b;
  b.benchmark_name = "quote\" backslash\\ newline\n tab\t cr\r unicode \xce\xb1 end";
  b.force_method = static_cast<ForceMethod>(7);
  b.metrics = {{"inf", std::numeric_limits<double>::infinity()}, {"nan", std::nan("")}, {"key \"q\"", 1.0}};
  records.push_back(b);
  BenchmarkRunRecord c;
  c.benchmark_name = "empty";
  c.force_method = ForceMethod::SPATIAL_HASH;
  records.push_back(c);
  std::printf("%s\n", serializeBenchmarkRunRecords(records).c_str());
  std::printf("%s\n", serializeBenchmarkRunRecord(a).c_str());
  return 0;
}
