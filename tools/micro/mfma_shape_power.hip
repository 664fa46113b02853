// Micro-experiment: sustained MFMA rate of the two fp16 shapes on a power-limited gfx950, registers only (no LDS, no HBM).
// 256 blocks x 8 waves (2 per SIMD), each wave holds a 128 x 64 accumulator tile (128 fp32 VGPRs) and loops over
// k-steps re-using register operands filled with random (or zero) data.  Question: does v_mfma_f32_32x32x16_f16 (half the
// operand-register reads per flop) sustain more FLOP/s than v_mfma_f32_16x16x32_f16 at the card's power limit?
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape_power.hip -o /tmp/mfma_shape_power ; run: /tmp/mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void k(const h8* __restrict__ src, float* out, int iters) {
  const int tid = threadIdx.x;
  h8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = src[(tid * 12 + i) & 4095];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = src[(tid * 12 + 8 + i) & 4095];
  float s = 0.f;
  if (SHAPE == 16) {
    f4 acc[8][4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    f16v acc[4][2] = {};
    for (int it = 0; it < iters; ++it) {
      // same flops per iteration: 128 x 64 x 32 -> 4 x 2 tiles x 2 k-steps of 16
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i * 2 + kk], b[j * 2 + kk], acc[i][j], 0, 0, 0);
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  }
  out[blockIdx.x * 512 + tid] = s;
}

template <int SHAPE>
static void run(const char* name, const h8* src, float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<SHAPE><<<256, 512>>>(src, out, iters);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<SHAPE><<<256, 512>>>(src, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
  const double fl = 2.0 * 128 * 64 * 32 * (double)iters * 8 * 256 * reps;
  printf("%-28s %8.3f ms/launch  %7.0f TFLOP/s\n", name, ms / reps, fl / (ms * 1e-3) * 1e-12);
}

int main() {
  std::vector<_Float16> h(4096 * 8);
  srand(1);
  h8* src; float* out;
  hipMalloc(&src, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;   // ~ 0.6 s per shape of sustained load
  for (int mode = 0; mode < 2; ++mode) {
    for (auto& v : h) v = mode ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const char* d = mode ? "zeros " : "random";
    char n[64];
    snprintf(n, 64, "16x16x32 f16 %s", d); run<16>(n, src, out, iters);
    snprintf(n, 64, "32x32x16 f16 %s", d); run<32>(n, src, out, iters);
    snprintf(n, 64, "16x16x32 f16 %s (again)", d); run<16>(n, src, out, iters);
  }
  return 0;
}
