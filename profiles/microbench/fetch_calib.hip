// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for the access widths libjmhip uses.
// Each kernel streams the same 256 MiB buffer (larger than the 256 MiB... L3 is bypassed by a second, different buffer) once.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ;  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o t -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void read_b4(const uint32_t *p, size_t n, uint32_t *out) {      // 4 bytes per lane, coalesced
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
  if (acc == 0x12345678u) *out = acc;
}
__global__ void read_b16(const uint4 *p, size_t n, uint32_t *out) {        // 16 bytes per lane, coalesced
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *out = acc;
}
struct __attribute__((packed)) u32u { uint32_t v; };
__global__ void read_b4_unaligned(const uint8_t *p, size_t n, uint32_t *out) {   // 4 bytes per lane at byte offset 1 (the ME window loads)
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= ((const u32u *)(p + 1 + 4 * i))->v;
  if (acc == 0x12345678u) *out = acc;
}
__global__ void read_b1(const uint8_t *p, size_t n, uint32_t *out) {       // 1 byte per lane
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
  if (acc == 0x12345678u) *out = acc;
}
int main() {
  const size_t bytes = 512ull << 20;
  uint8_t *a, *b; uint32_t *out;
  hipMalloc(&a, bytes + 64); hipMalloc(&b, bytes + 64); hipMalloc(&out, 4);
  hipMemset(a, 1, bytes + 64); hipMemset(b, 2, bytes + 64);
  for (int rep = 0; rep < 2; rep++) {
    uint8_t *p = rep ? b : a;
    read_b4<<<4096, 256>>>((const uint32_t *)p, bytes / 4, out);
    read_b16<<<4096, 256>>>((const uint4 *)p, bytes / 16, out);
    read_b4_unaligned<<<4096, 256>>>(p, bytes / 4 - 1, out);
    read_b1<<<4096, 256>>>(p, bytes / 8, out);          // 64 MiB
  }
  hipDeviceSynchronize();
  printf("bytes per kernel: b4 %zu  b16 %zu  b4_unaligned %zu  b1 %zu\n", bytes, bytes, bytes, bytes / 8);
  return 0;
}
