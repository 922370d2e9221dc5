// write_calib.hip -- what rocprofv3's WRITE_SIZE counts on gfx950 for the store forms the post stage of the macroblock pipeline uses (mbpipe_post.inc): a 16-byte piece of
// each 128-byte line written by a different lane group (as neighbouring macroblocks' workgroups write neighbouring 16-byte pieces of a plane's row at different times),
// write-through (sc1) or plain, against whole lines.  Every kernel stores the same number of payload bytes (64 MiB).
//   hipcc --offload-arch=gfx950 -O3 -o write_calib write_calib.hip ;  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -o t -- ./write_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define N_PIECES (4u << 20)                       // 4 Mi pieces of 16 bytes = 64 MiB of payload

// piece i lands at byte 128 * i: one 16-byte piece per 128-byte line (the rest of the line is never written)
__global__ void w16_sc1_sparse(uint8_t *p) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < N_PIECES / 8; i += gridDim.x * blockDim.x) { v4u v = {i, i, i, i}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(i * 128u), 0, 16); }
}
// the same pieces, every line completed piece by piece by DIFFERENT passes of the kernel (pass k writes piece k of every line): what eight macroblocks in a row do to a plane's rows
__global__ void w16_sc1_piecewise(uint8_t *p, int k) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < N_PIECES / 8; i += gridDim.x * blockDim.x) { v4u v = {i, i, i, i}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(i * 128u + 16u * k), 0, 16); }
}
__global__ void w16_plain_piecewise(uint8_t *p, int k) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < N_PIECES / 8; i += gridDim.x * blockDim.x) { v4u v = {i, i, i, i}; *(v4u *)(p + (size_t)i * 128u + 16u * k) = v; }
}
// 4- and 8-byte write-through stores (the edge records, the side information, the picture's rows), one per 128-byte line
__global__ void w8_sc1_sparse(uint8_t *p) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < N_PIECES / 8; i += gridDim.x * blockDim.x) __hip_atomic_store((unsigned long long *)(p + (size_t)i * 128u), (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void w4_sc1_sparse(uint8_t *p) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < N_PIECES / 8; i += gridDim.x * blockDim.x) __hip_atomic_store((unsigned *)(p + (size_t)i * 128u), i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// whole lines: contiguous 16-byte stores, write-through and plain
__global__ void w16_sc1_dense(uint8_t *p) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < N_PIECES; i += gridDim.x * blockDim.x) { v4u v = {i, i, i, i}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(i * 16u), 0, 16); }
}
__global__ void w16_plain_dense(uint8_t *p) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < N_PIECES; i += gridDim.x * blockDim.x) { v4u v = {i, i, i, i}; *(v4u *)(p + (size_t)i * 16u) = v; }
}
int main() {
  uint8_t *a;
  hipMalloc(&a, (size_t)N_PIECES * 16u + 256);
  hipMemset(a, 0, (size_t)N_PIECES * 16u + 256);
  hipDeviceSynchronize();
  w16_sc1_dense<<<2048, 256>>>(a); hipDeviceSynchronize();                 // 64 MiB payload, whole lines
  w16_plain_dense<<<2048, 256>>>(a); hipDeviceSynchronize();
  w16_sc1_sparse<<<2048, 256>>>(a); hipDeviceSynchronize();                // 8 MiB payload: one piece per line
  for (int k = 0; k < 8; k++) { w16_sc1_piecewise<<<2048, 256>>>(a, k); hipDeviceSynchronize(); }     // 8 x 8 MiB
  for (int k = 0; k < 8; k++) { w16_plain_piecewise<<<2048, 256>>>(a, k); hipDeviceSynchronize(); }
  w8_sc1_sparse<<<2048, 256>>>(a); hipDeviceSynchronize();                 // 4 MiB payload
  w4_sc1_sparse<<<2048, 256>>>(a); hipDeviceSynchronize();                 // 2 MiB payload
  printf("done\n");
  return 0;
}
