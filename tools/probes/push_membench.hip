// Memory-phase microbenchmark of the fused push of qr_factor_kernel<float, 4, true, 8, true> (round 6): WHAT binds the push?
//
// The level-0 block of the metric workload (packed) reads 2 x 8 core slices C[:, i, :] (64 rows of 256 B, rows 16 KB apart) =
// 256 KB per 512-thread block, two blocks per CU, then runs ~68 k cycles of serial panel chain without reading HBM.  This file
// reproduces that access pattern and duty cycle with three ways of moving the bytes:
//   mode 0  the kernel's: global_load_dword in MFMA B-operand layout (lane (g, cl): row 4 ks + g, column 16 tn + cl; one
//           instruction = 4 rows x 64 B), two K groups (2 x 16 loads per lane) in flight
//   mode 1  global_load_dwordx4 (lane (r, c): row 4 ks + r, columns 4 c .. 4 c + 3; one instruction = 4 rows x 256 B), the same
//           bytes in flight
//   mode 2  global_load_lds_dwordx4 into a per-wave LDS ring of SLOTS x 1 KB (source-swizzled so that the B-operand ds_read_b32
//           are conflict-free), read back with 4 ds_read_b32 per K step
// and a `spin` of S kilo-cycles after the memory phase (s_sleep: the panel chain's HBM silence).  `stagger`: workgroups 256 .. 511
// of the launch wait that many kilo-cycles before they start (the second resident block of every CU, if dispatch is round-robin).
//
//   hipcc --offload-arch=gfx950 -O3 -o push_membench push_membench.hip && ./push_membench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kI = 64, kN = 64, kR = 64;          // core [kR][kI][kN]
constexpr int kItem = kR * kI * kN;              // floats per item
constexpr int LDS_PAD = 72 * 1024;               // (mode 2 with SLOTS = 8 uses 64 KB of it)               // static LDS per block: exactly two blocks per CU

__device__ __forceinline__ void spin_kc(int kc) {
  if (kc <= 0) return;
  const long long t0 = clock64();
  while (clock64() - t0 < (long long)kc * 1024) __builtin_amdgcn_s_sleep(16);
}

// one LDS-DMA: every lane's 16 bytes at `gsrc` land at `lds_dst` (wave-uniform) + 16 * lane
__device__ __forceinline__ void glds16(const float* gsrc, float* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
#endif
}

template <int MODE, int SLOTS>
__global__ __launch_bounds__(512, 4) void push_kernel(const float* cores, float* out, long long* cyc,
                                                      int spin, int stagger, int reps) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_PAD / 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
  const long long id = blockIdx.x;
  if (stagger > 0 && id >= 256 && id < 512) spin_kc(stagger);
  float acc = 0.f;
  long long push_cyc = 0;
  for (int rep = 0; rep < reps; ++rep) {
    // (reps > 1: a persistent workgroup walking `reps` work items gridDim.x apart)
    const long long wid = id + (long long)rep * gridDim.x;
    const long long item = wid >> 2;
    const int b = (int)(wid & 3);
    const float* base = cores + item * (long long)kItem;
    const long long t0 = clock64();
    if constexpr (MODE == 0) {
      const int g = lane >> 4, cl = lane & 15;
      float bvA[4][4], bvB[4][4];
      for (int pass = 0; pass < 2; ++pass) {
        const int im = 8 * (b + 4 * pass) + wave;
        const float* sl = base + im * kN;
        auto load_group = [&](int grp, float (&bv)[4][4]) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) bv[kk][tn] = sl[(size_t)((grp * 4 + kk) * 4 + g) * (kI * kN) + tn * 16 + cl];
        };
        auto use_group = [&](const float (&bv)[4][4]) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc += bv[kk][tn];
          asm volatile("" : "+v"(acc));   // the group's values are consumed HERE (otherwise the adds sink below every load)
        };
        // (compiler fences: without them hipcc hoists all 64 loads of both passes to the top -- the kernel's MFMAs and register
        // budget keep two groups in flight)
        load_group(0, bvA);
        load_group(1, bvB);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        use_group(bvA);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        load_group(2, bvA);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        use_group(bvB);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        load_group(3, bvB);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        use_group(bvA);
        use_group(bvB);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (MODE == 3) {
      // mode 0's loads with SLOTS K GROUPS (16 loads per lane each) in flight instead of two
      const int g = lane >> 4, cl = lane & 15;
      float bv[SLOTS][4][4];
      auto sl_of = [&](int pass) { return base + (8 * (b + 4 * pass) + wave) * kN; };
      auto load_group = [&](int gi, float (&q)[4][4]) {   // gi = 0 .. 7 over both passes
        const float* sl = sl_of(gi >> 2);
        const int grp = gi & 3;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) q[kk][tn] = sl[(size_t)((grp * 4 + kk) * 4 + g) * (kI * kN) + tn * 16 + cl];
      };
#pragma unroll
      for (int gi = 0; gi < SLOTS; ++gi) { load_group(gi, bv[gi]); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) acc += bv[gi % SLOTS][kk][tn];
        asm volatile("" : "+v"(acc));
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        if (gi + SLOTS < 8) { load_group(gi + SLOTS, bv[gi % SLOTS]); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
      }
    } else if constexpr (MODE == 1) {
      const int r = lane >> 4, c = lane & 15;
      f4 bvA[4], bvB[4];
      for (int pass = 0; pass < 2; ++pass) {
        const int im = 8 * (b + 4 * pass) + wave;
        const float* sl = base + im * kN;
        auto load_group = [&](int grp, f4 (&bv)[4]) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) bv[kk] = *reinterpret_cast<const f4*>(sl + (size_t)((grp * 4 + kk) * 4 + r) * (kI * kN) + 4 * c);
        };
        auto use_group = [&](const f4 (&bv)[4]) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc += bv[kk].x + bv[kk].y + bv[kk].z + bv[kk].w;
          asm volatile("" : "+v"(acc));
        };
        // (twice the groups in flight per instruction count: 2 groups = 8 x 1 KB per wave, as mode 0)
        // (compiler fences: without them hipcc hoists all 64 loads of both passes to the top -- the kernel's MFMAs and register
        // budget keep two groups in flight)
        load_group(0, bvA);
        load_group(1, bvB);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        use_group(bvA);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        load_group(2, bvA);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        use_group(bvB);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        load_group(3, bvB);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        use_group(bvA);
        use_group(bvB);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // ring of SLOTS x 1 KB per wave; K step ks (0 .. 31 over both passes) -> slot ks % SLOTS
      float* const ring = lds + wave * (SLOTS * 256);
      const int p = lane;                                   // 16-byte unit this lane's DMA element lands in
      const int srow = p >> 4, sq = (p & 15) ^ (4 * ((p >> 4) & 1));   // source (row in the K step, column quad): swizzled
      const int g = lane >> 4, cl = lane & 15;
      auto src = [&](int ks) -> const float* {
        const int pass = ks >> 4, k16 = ks & 15;
        const int im = 8 * (b + 4 * pass) + wave;
        return base + im * kN + (size_t)(k16 * 4 + srow) * (kI * kN) + 4 * sq;
      };
      auto issue = [&](int ks) {
        glds16(src(ks), ring + (ks % SLOTS) * 256);
      };
#pragma unroll
      for (int ks = 0; ks < SLOTS; ++ks) issue(ks);
#pragma unroll 1
      for (int ks0 = 0; ks0 < 32; ks0 += SLOTS) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          const int ks = ks0 + s;
          // the oldest DMA has landed when at most SLOTS - 1 are outstanding (tail: fewer were issued)
          const int left = 32 - 1 - ks;   // DMAs issued after this one
          if (left >= SLOTS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS - 1) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const float* slot = ring + s * 256;
          float v[4];
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) {
            const int q = tn * 4 + (cl >> 2);
            v[tn] = slot[4 * (16 * g + (q ^ (4 * (g & 1)))) + (cl & 3)];
          }
          acc += (v[0] + v[1]) + (v[2] + v[3]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the slot is refilled
          if (ks + SLOTS < 32) issue(ks + SLOTS);
        }
      }
    }
    // the loads must have landed before the phase counts as over
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    push_cyc += clock64() - t0;
    spin_kc(spin);
  }
  // (every mode touches the static LDS array: otherwise hipcc drops it and modes 0 / 1 run four blocks per CU, mode 2 two)
  lds[LDS_PAD / 4 - 512 + tid] = acc;
  __syncthreads();
  out[blockIdx.x * 512 + tid] = lds[LDS_PAD / 4 - 512 + ((tid + 1) & 511)];
  if (tid == 0) cyc[blockIdx.x] = push_cyc;
}

template <int MODE, int SLOTS>
static void run(const char* name, const float* cores, float* out, long long* cyc, int nblocks, int spin, int stagger, int reps, double clk_ghz) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = nblocks / reps;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((push_kernel<MODE, SLOTS>), dim3(grid), dim3(512), 0, 0, cores, out, cyc, spin, stagger, reps);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((push_kernel<MODE, SLOTS>), dim3(grid), dim3(512), 0, 0, cores, out, cyc, spin, stagger, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  std::vector<long long> h(grid);
  CK(hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
  for (auto& x : h) x /= reps;
  std::sort(h.begin(), h.end());
  const double bytes = (double)nblocks * 256.0 * 1024.0;
  double mean = 0;
  for (auto x : h) mean += (double)x;
  mean /= grid;
  // per-block memory phase in cycles (s_memtime ticks at 100 MHz on gfx9: scale by the shader clock to compare with stamps)
  printf("%-34s spin %3dk stagger %3dk reps %2d: %7.3f ms  %6.2f TB/s  push ticks/block p10 %lld  p50 %lld  p90 %lld  mean %.0f\n", name, spin, stagger, reps, best,
         bytes / best / 1e9, h[grid / 10], h[grid / 2], h[(grid * 9) / 10], mean);
  (void)clk_ghz;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096;
  const int nblocks = B * 4;
  float *cores, *out;
  long long* cyc;
  CK(hipMalloc(&cores, (size_t)B * kItem * 4));
  CK(hipMalloc(&out, (size_t)nblocks * 512 * 4));
  CK(hipMalloc(&cyc, (size_t)nblocks * 8));
  CK(hipMemset(cores, 0, (size_t)B * kItem * 4));
  printf("# B = %d items (%.2f GB of cores), %d working blocks of 512 threads, 256 KB read per block, 2 blocks per CU (72 KB static LDS)\n", B,
         (double)B * kItem * 4 / 1e9, nblocks);
  // clock64 = s_memtime: report its rate once
  {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((push_kernel<0, 4>), dim3(256), dim3(512), 0, 0, cores, out, cyc, 1000, 0, 1);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("# a 1000 k-tick spin takes %.3f ms -> clock64 ticks at %.1f MHz\n", ms, 1024.0 * 1000 / ms / 1e3);
  }
  // in-flight depth (bytes per block in flight = groups x 32 KB), loaded chip and one block per CU on an idle chip
  for (int nb : {nblocks, 256}) {
    run<3, 1>("3: dword, 1 group in flight", cores, out, cyc, nb, 68, 0, 1, 2.4);
    run<3, 2>("3: dword, 2 groups in flight", cores, out, cyc, nb, 68, 0, 1, 2.4);
    run<3, 4>("3: dword, 4 groups in flight", cores, out, cyc, nb, 68, 0, 1, 2.4);
    run<3, 8>("3: dword, 8 groups in flight", cores, out, cyc, nb, 68, 0, 1, 2.4);
    run<2, 4>("2: LDS-DMA ring 4 KB/wave", cores, out, cyc, nb, 68, 0, 1, 2.4);
    run<2, 8>("2: LDS-DMA ring 8 KB/wave", cores, out, cyc, nb, 68, 0, 1, 2.4);
  }
  const int spins[] = {0, 68};
  for (int spin : spins) {
    run<0, 4>("0: dword, B-operand layout", cores, out, cyc, nblocks, spin, 0, 1, 2.4);
    run<1, 4>("1: dwordx4 to registers", cores, out, cyc, nblocks, spin, 0, 1, 2.4);
    run<2, 4>("2: LDS-DMA x4, ring 4 KB/wave", cores, out, cyc, nblocks, spin, 0, 1, 2.4);
    run<2, 8>("2: LDS-DMA x4, ring 8 KB/wave", cores, out, cyc, nblocks, spin, 0, 1, 2.4);
  }
  // de-phasing the two resident blocks of a CU
  for (int st : {20, 45}) {
    run<0, 4>("0: dword, staggered start", cores, out, cyc, nblocks, 68, st, 1, 2.4);
    run<2, 4>("2: LDS-DMA ring 4 KB, staggered", cores, out, cyc, nblocks, 68, st, 1, 2.4);
  }
  // persistent workgroups (512 resident blocks walking 32 work items each)
  run<0, 4>("0: dword, persistent", cores, out, cyc, nblocks, 68, 0, 32, 2.4);
  run<0, 4>("0: dword, persistent, staggered", cores, out, cyc, nblocks, 68, 45, 32, 2.4);
  run<2, 4>("2: LDS-DMA 4 KB, persistent, stag.", cores, out, cyc, nblocks, 68, 45, 32, 2.4);
  return 0;
}
