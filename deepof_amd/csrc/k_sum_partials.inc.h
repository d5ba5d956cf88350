// k_sum_partials_multi's workgroup body (one output value per workgroup: strided partial sums + a fixed-shape LDS tree),
// shared by k_reduce.hip's kernel and k_rnn.hip's k_step_finalize.  `red`: 256 floats of LDS.
#pragma once
__device__ __forceinline__ void dof_sum_partials_multi_body(const DofSumJobs& J, int v, int accumulate, float* red) {
  int j = 0;
  while (j + 1 < J.n && v >= J.nv[j]) {
    v -= J.nv[j];
    ++j;
  }
  const float* __restrict__ partial = J.partial[j];
  const int64_t nblk = J.nblk[j];
  const int nv = J.nv[j];
  float acc = 0.0f;
  for (int64_t b = threadIdx.x; b < nblk; b += 256) acc += partial[b * nv + v];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) J.out[j][v] = accumulate ? J.out[j][v] + red[0] : red[0];
}
