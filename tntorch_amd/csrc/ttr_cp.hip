// CP-ALS building blocks (tensor.py:279-394) and the TT x TT core product: the
// Khatri-Rao contraction that turns the
// partially contracted tensor of a fused MTTKRP into the next one, and the Hadamard product of the R x R
// Gram matrices.  HBM-streaming kernels: T is read exactly once, coalesced; no Khatri-Rao matrix and no
// permuted unfolding copy is ever materialised.
#include "ttr_common.h"

namespace ttr {

// out[p, x] = sum_j T[p, j, x] * B[j, x % R],  x in [0, inner = Q*R).
// Workgroup = (p, x-chunk of XC columns); the 256 threads form JS = 256 / XC slices over j that are reduced
// through LDS at the end.  Consecutive threads read consecutive x: every wave load is one contiguous segment.
template <typename T>
__global__ __launch_bounds__(kThreads) void krp_contract_kernel(const T* __restrict__ Tn, const T* __restrict__ B,
                                                                 T* __restrict__ out, int64_t J, int64_t inner,
                                                                 int64_t R, int64_t ldb, int XC, int JS) {
  __shared__ T red[kThreads];
  const int tid = threadIdx.x;
  const int xl = tid % XC, js = tid / XC;
  const int64_t p = blockIdx.y;
  const int64_t x = (int64_t)blockIdx.x * XC + xl;
  const bool act = js < JS && x < inner;
  T acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  if (act) {
    const int64_t r = x % R;
    const T* __restrict__ tp = Tn + p * J * inner + x;
    int64_t j = js;
    const int64_t step = JS;
    for (; j + 3 * step < J; j += 4 * step) {  // four independent loads in flight per thread
      const T t0 = tp[j * inner], t1 = tp[(j + step) * inner], t2 = tp[(j + 2 * step) * inner],
              t3 = tp[(j + 3 * step) * inner];
      acc0 += t0 * B[j * ldb + r];
      acc1 += t1 * B[(j + step) * ldb + r];
      acc2 += t2 * B[(j + 2 * step) * ldb + r];
      acc3 += t3 * B[(j + 3 * step) * ldb + r];
    }
    for (; j < J; j += step) acc0 += tp[j * inner] * B[j * ldb + r];
  }
  red[tid] = (acc0 + acc1) + (acc2 + acc3);
  __syncthreads();
  if (js == 0 && x < inner) {
    T s = red[xl];
    for (int k = 1; k < JS; ++k) s += red[k * XC + xl];
    out[p * inner + x] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void hadamard_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                             T* __restrict__ out, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += (int64_t)gridDim.x * kThreads)
    out[i] = a[i] * b[i];
}

// out[b, r1*S1 + s1, i, r2*S2 + s2] = a[b, r1, i, r2] * c[b, s1, i, s2]  (slice-wise Kronecker product of two TT cores).
// One workgroup per output row (b, r1, s1, i): the row index is decoded once per workgroup, the threads walk the
// contiguous R2*S2 columns (write-coalesced; the a / c rows they read are R2 and S2 elements long and stay in L1).
template <typename T>
__global__ __launch_bounds__(kThreads) void core_kron_kernel(const T* __restrict__ a, const T* __restrict__ c,
                                                              T* __restrict__ out, int R1, int S1, int I, int R2,
                                                              int S2) {
  int64_t t = blockIdx.x;
  const int i = (int)(t % I); t /= I;
  const int s1 = (int)(t % S1); t /= S1;
  const int r1 = (int)(t % R1);
  const int64_t b = t / R1;
  const T* __restrict__ ar = a + ((b * R1 + r1) * I + i) * (int64_t)R2;
  const T* __restrict__ cr = c + ((b * S1 + s1) * I + i) * (int64_t)S2;
  T* __restrict__ orow = out + (int64_t)blockIdx.x * R2 * S2;
  const int w = R2 * S2;
  for (int col = threadIdx.x; col < w; col += kThreads) {
    const int r2 = col / S2, s2 = col - r2 * S2;
    orow[col] = ar[r2] * cr[s2];
  }
}

int core_kron_dispatch(int dtype, int64_t B, int64_t R1, int64_t S1, int64_t I, int64_t R2, int64_t S2, const void* a,
                       const void* c, void* out, hipStream_t stream) {
  const int64_t rows = B * R1 * S1 * I;
  TTR_REQUIRE(rows <= 2147483647LL && R2 * S2 <= 2147483647LL, TTR_E_UNSUPPORTED, "ttr_core_kron: core too large");
  ProfScope prof(TTR_PROF_MISC, stream);
  if (dtype == TTR_F32)
    hipLaunchKernelGGL(core_kron_kernel<float>, dim3((unsigned)rows), dim3(kThreads), 0, stream, (const float*)a,
                       (const float*)c, (float*)out, (int)R1, (int)S1, (int)I, (int)R2, (int)S2);
  else
    hipLaunchKernelGGL(core_kron_kernel<double>, dim3((unsigned)rows), dim3(kThreads), 0, stream, (const double*)a,
                       (const double*)c, (double*)out, (int)R1, (int)S1, (int)I, (int)R2, (int)S2);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int krp_contract_dispatch(int dtype, int64_t P, int64_t J, int64_t Q, int64_t R, const void* Tn, const void* B,
                          int64_t ldb, void* out, hipStream_t stream) {
  const int64_t inner = Q * R;
  int XC = inner >= kThreads ? kThreads : (int)inner;
  int JS = kThreads / XC;
  if (JS > J) JS = (int)J;
  const int64_t chunks = ceil_div(inner, XC);
  TTR_REQUIRE(P <= 65535 || chunks == 1, TTR_E_UNSUPPORTED, "ttr_krp_contract: P = %lld > 65535 with inner > 256",
              (long long)P);
  dim3 grid, block(kThreads);
  // grid.y is limited to 65535: for inner <= 256 (one chunk) p goes to grid.x instead
  ProfScope prof(TTR_PROF_MISC, stream);
  if (chunks == 1) {
    // p on x: emulate blockIdx.y = p by launching with (1, P) when P fits, else fold through pointer offsets
    int64_t done = 0;
    while (done < P) {
      const int64_t n = (P - done) < 65535 ? (P - done) : 65535;
      grid = dim3(1, (unsigned)n);
      const int64_t off = done * J * inner, ooff = done * inner;
      if (dtype == TTR_F32)
        hipLaunchKernelGGL(krp_contract_kernel<float>, grid, block, 0, stream, (const float*)Tn + off, (const float*)B,
                           (float*)out + ooff, J, inner, R, ldb, XC, JS);
      else
        hipLaunchKernelGGL(krp_contract_kernel<double>, grid, block, 0, stream, (const double*)Tn + off,
                           (const double*)B, (double*)out + ooff, J, inner, R, ldb, XC, JS);
      done += n;
    }
  } else {
    grid = dim3((unsigned)chunks, (unsigned)P);
    if (dtype == TTR_F32)
      hipLaunchKernelGGL(krp_contract_kernel<float>, grid, block, 0, stream, (const float*)Tn, (const float*)B,
                         (float*)out, J, inner, R, ldb, XC, JS);
    else
      hipLaunchKernelGGL(krp_contract_kernel<double>, grid, block, 0, stream, (const double*)Tn, (const double*)B,
                         (double*)out, J, inner, R, ldb, XC, JS);
  }
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

int hadamard_dispatch(int dtype, int64_t count, const void* a, const void* b, void* out, hipStream_t stream) {
  int64_t gx = ceil_div(count, kThreads);
  if (gx > 4096) gx = 4096;
  ProfScope prof(TTR_PROF_MISC, stream);
  if (dtype == TTR_F32)
    hipLaunchKernelGGL(hadamard_kernel<float>, dim3((unsigned)gx), dim3(kThreads), 0, stream, (const float*)a,
                       (const float*)b, (float*)out, count);
  else
    hipLaunchKernelGGL(hadamard_kernel<double>, dim3((unsigned)gx), dim3(kThreads), 0, stream, (const double*)a,
                       (const double*)b, (double*)out, count);
  TTR_HIP_CHECK(hipGetLastError());
  return TTR_OK;
}

}  // namespace ttr
