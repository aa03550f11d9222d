#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* y) {
  __shared__ __attribute__((aligned(16))) short lds[64*32];
  for (int i = threadIdx.x; i < 64*32; i += 64) lds[i] = (short)((i/32)*100 + (i%32));
  __syncthreads();
  int l = threadIdx.x;
  int i = l & 15, q = i >> 2, p = i & 3, g = l >> 4;
  int colbase = 16*(g&1), kbase = 8*(g>>1);
  short* addr = &lds[(kbase + q) * 32 + colbase + 4 * p];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int j = 0; j < 4; ++j) y[l*4+j] = (float)v[j];
}
int main(){ float* d; hipMalloc(&d, 256*4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); float h[256]; hipMemcpy(h,d,1024,hipMemcpyDeviceToHost);
 for(int l=0;l<64;++l){ printf("lane %2d:",l); for(int j=0;j<4;++j) printf(" %5.0f",h[l*4+j]); printf("\n"); } return 0; }
