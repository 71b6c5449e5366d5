// Host harness for the product's model compiler (csrc/model.cpp + csrc/xml_min.hpp) - the code that parses
// caller-named files. Built by tests/test_host_sanitize.py with -fsanitize=address,undefined; no HIP, no stubs:
// model.cpp has no device dependency.
//
//   model_harness [--dae <collisions_dir>] [--primitives] <file.urdf> ...
//
// For every file: "<path>\t<code>\t<bodies or message>". Exit code 0 unless the loader let anything but its own
// std::runtime_error escape (a sanitizer report aborts the process on its own).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>

#include "../../trex-gym_amd/csrc/model.hpp"

int main(int argc, char **argv) {
  const char *dae = nullptr;
  bool primitives = false;
  int rc = 0;
  for (int i = 1; i < argc; i++) {
    if (!std::strcmp(argv[i], "--dae") && i + 1 < argc) { dae = argv[++i]; continue; }
    if (!std::strcmp(argv[i], "--primitives")) { primitives = true; continue; }
    int code = 12345;
    try {
      trex::HostModel m = trex::load_model(argv[i], dae, &code);
      if (primitives && !m.hull_xyz.empty()) trex::use_primitive_collision(m, 0.15, 4, 4);
      // a model that loads is FINITE: every number the kernels will read
      bool finite = std::isfinite(m.total_mass);
      for (int b = 0; b < m.nb; b++) {
        finite = finite && std::isfinite(m.mass[b]) && m.mass[b] > 0 && std::isfinite(m.q_lower[b]) && std::isfinite(m.q_upper[b]);
        for (double v : m.inertia[b]) finite = finite && std::isfinite(v);
        finite = finite && std::isfinite(m.com[b].x + m.com[b].y + m.com[b].z) && std::isfinite(m.joint_axis[b].x + m.joint_axis[b].y + m.joint_axis[b].z);
      }
      for (auto &p : m.hull_xyz) finite = finite && std::isfinite(p.x + p.y + p.z);
      std::printf("%s\t%d\t%d bodies, %zu hull points%s\n", argv[i], code, m.nb, m.hull_xyz.size(), finite ? "" : " NON-FINITE");
      if (!finite || code != 0) rc = 1;
    } catch (const std::runtime_error &e) {
      std::printf("%s\t%d\t%s\n", argv[i], code, e.what());
      if (code >= 0 || code == 12345) rc = 1;     // a refusal carries one of the negative TREX_E_* codes
    } catch (...) {
      std::printf("%s\t%d\tFOREIGN EXCEPTION\n", argv[i], code);
      rc = 1;
    }
  }
  return rc;
}
