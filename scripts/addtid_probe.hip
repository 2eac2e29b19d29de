#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float s[];
__global__ void k(float *o)
{
   // each wave parks its value in row 1 of its own 64-float column block
   float v = (float)threadIdx.x * 2.0f + 1.0f;
   int wave_byte = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u) * 4);
   unsigned keep;
   asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:1024\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(v), "s"(wave_byte) : "memory");
   __syncthreads();
   o[threadIdx.x] = s[256 + threadIdx.x]; // plain read of [row 1][thread]
   float r;
   asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tds_read_addtid_b32 %1 offset:1024\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %0" : "=&s"(keep), "=v"(r) : "s"(wave_byte) : "memory");
   o[256 + threadIdx.x] = r;
}
int main()
{
   float *d, h[512];
   hipMalloc(&d, sizeof h);
   hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d);
   hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
   int bad = 0;
   for (int i = 0; i < 256; ++i) if (h[i] != i * 2.0f + 1.0f || h[256 + i] != i * 2.0f + 1.0f) ++bad;
   printf("addtid: %d mismatches (h[65]=%g h[321]=%g)\n", bad, h[65], h[321]);
   return bad != 0;
}
