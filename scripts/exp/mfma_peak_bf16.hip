#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template<int NACC>
__global__ __launch_bounds__(256) void peak(float* out, const bf16x8* in, int iters){
  f32x16 acc[NACC];
  for(int i=0;i<NACC;i++) for(int r=0;r<16;r++) acc[i][r]=0.f;
  bf16x8 a = in[threadIdx.x], b = in[threadIdx.x+256];
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int u=0;u<8;u++)
#pragma unroll
      for(int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,acc[i],0,0,0);
  }
  float s=0; for(int i=0;i<NACC;i++) for(int r=0;r<16;r++) s+=acc[i][r];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
extern "C" void run(float* out, const void* in, int grid, int iters, int nacc, hipStream_t st){
  if(nacc==2) hipLaunchKernelGGL(peak<2>,dim3(grid),dim3(256),0,st,out,(const bf16x8*)in,iters);
  if(nacc==4) hipLaunchKernelGGL(peak<4>,dim3(grid),dim3(256),0,st,out,(const bf16x8*)in,iters);
}
