// C++ caller of the C-ABI (include/trex_batch.h) with plain hipMalloc buffers: no Python, no torch.
//   ./trex_capi_example <urdf> [num_envs] [steps]
// Loads the model, resets, steps with a fixed action, prints obs/reward of env 0 and the throughput.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/trex_batch.h"

#define CHECK(x)                                                               \
  do {                                                                         \
    if ((x) != 0) { std::fprintf(stderr, "%s failed: %s\n", #x, trex_last_error()); return 1; } \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s <urdf> [num_envs] [steps]\n", argv[0]); return 2; }
  const int n = argc > 2 ? std::atoi(argv[2]) : 4096, steps = argc > 3 ? std::atoi(argv[3]) : 200;
  TrexModel *model = nullptr;
  TrexBatch *batch = nullptr;
  CHECK(trex_model_load(argv[1], nullptr, &model));
  const int J = trex_model_num_joints(model);
  std::printf("model: %d bodies, %d joints (%d URDF joints), %d hull vertices, mass %.2f kg\n", trex_model_num_bodies(model), J,
              trex_model_num_urdf_joints(model), trex_model_num_hull_vertices(model), trex_model_total_mass(model, 1));
  CHECK(trex_batch_create(model, n, 0, &batch));
  float *act, *obs, *rew;
  uint8_t *done;
  if (hipMalloc(&act, sizeof(float) * n * J) || hipMalloc(&obs, sizeof(float) * n * 3 * J) || hipMalloc(&rew, sizeof(float) * n) ||
      hipMalloc(&done, n)) { std::fprintf(stderr, "hipMalloc failed\n"); return 1; }
  std::vector<float> a(size_t(n) * J, 0.f);
  for (int e = 0; e < n; e++)
    for (int k = 0; k < J; k++) {
      double lo, hi;
      trex_model_joint_info(model, k, nullptr, nullptr, &lo, &hi);
      a[size_t(e) * J + k] = float(lo + (hi - lo) * ((e * 31 + k * 17) % 97) / 96.0);
    }
  hipMemcpy(act, a.data(), a.size() * sizeof(float), hipMemcpyHostToDevice);
  hipStream_t stream;
  hipStreamCreate(&stream);
  CHECK(trex_batch_reset(batch, nullptr, obs, stream));
  {
    // A host pointer or a short buffer must come back as TREX_E_INVALID (it would fault the GPU otherwise);
    // the batch stays usable afterwards.
    std::vector<float> host_obs(size_t(n) * 3 * J);
    float *small = nullptr;
    if (hipMalloc(&small, 64)) return 1;
    const int e1 = trex_batch_step(batch, a.data(), obs, rew, done, nullptr, stream);        // host actions
    const int e2 = trex_batch_step(batch, act, host_obs.data(), rew, done, nullptr, stream);  // host obs
    const int e3 = trex_batch_step(batch, act, obs, small, done, nullptr, stream);            // reward buffer too short
    const int e4 = trex_batch_step_rows(batch, act, obs, 3 * J, nullptr, nullptr, stream);             // row stride < 3J + 2
    if (e1 != TREX_E_INVALID || e2 != TREX_E_INVALID || e3 != TREX_E_INVALID || e4 != TREX_E_INVALID) {
      std::fprintf(stderr, "bad buffers were not refused: %d %d %d %d\n", e1, e2, e3, e4);
      return 1;
    }
    std::printf("bad buffers refused: %s\n", trex_last_error());
    hipFree(small);
  }
  for (int t = 0; t < 20; t++) CHECK(trex_batch_step(batch, act, obs, rew, done, nullptr, stream));
  hipStreamSynchronize(stream);
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < steps; t++) CHECK(trex_batch_step(batch, act, obs, rew, done, nullptr, stream));
  hipStreamSynchronize(stream);
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::vector<float> o(3 * J);
  float r0;
  hipMemcpy(o.data(), obs, o.size() * sizeof(float), hipMemcpyDeviceToHost);
  hipMemcpy(&r0, rew, sizeof(float), hipMemcpyDeviceToHost);
  std::printf("env 0: q[0..2] = %.4f %.4f %.4f  reward %.3f\n", o[0], o[1], o[2], r0);
  std::printf("%d envs x %d steps in %.3f s = %.0f env-steps/s\n", n, steps, sec, double(n) * steps / sec);
  trex_batch_destroy(batch);
  trex_model_destroy(model);
  return 0;
}
