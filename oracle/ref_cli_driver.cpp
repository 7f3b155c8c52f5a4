// ref_cli_driver.cpp -- test infrastructure (oracle/): runs the REFERENCE's own parseAppCliOptions
// (compiled from /root/reference/src/core/app_cli.cpp by Makefile.ref; host only) on the given
// arguments and prints every option, or "error: <what()>" with exit code 1 -- so that tests can
// compare n-body_amd/cli.py with the real parser argument vector by argument vector.
#include <cstdio>
#include <exception>

#include "nbody/app_cli.hpp"

using namespace nbody;

int main(int argc, char** argv) {
  try {
    const AppCliOptions o = parseAppCliOptions(argc, argv);
    std::printf("particle_count %zu\nforce_method %d\ndt %.9g\nG %.9g\nsoftening %.9g\ntheta %.9g\n"
                "cell_size %.9g\ncutoff %.9g\nbenchmark_mode %d\nbenchmark_steps %zu\nbenchmark_output %s\n"
                "show_help %d\nexport_path %s\nexport_format %s\nimport_path %s\nlist_algorithms %d\n"
                "show_diagnostics %d\n",
                o.particle_count, static_cast<int>(o.force_method), o.dt, o.G, o.softening, o.barnes_hut_theta,
                o.spatial_hash_cell_size, o.spatial_hash_cutoff, o.benchmark_mode ? 1 : 0, o.benchmark_steps,
                o.benchmark_output_path.c_str(), o.show_help ? 1 : 0, o.export_path.c_str(),
                o.export_format.c_str(), o.import_path.c_str(), o.list_algorithms ? 1 : 0,
                o.show_diagnostics ? 1 : 0);
    return 0;
  } catch (const std::exception& e) {
    std::printf("error: %s\n", e.what());
    return 1;
  }
}
