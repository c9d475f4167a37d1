#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template<int NACC>
__global__ __launch_bounds__(256) void peak(float* out, int iters, float a, float b){
  f32x16 acc[NACC];
  for(int i=0;i<NACC;i++) for(int r=0;r<16;r++) acc[i][r]=0.f;
  float x=a+threadIdx.x, y=b;
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int u=0;u<8;u++)
#pragma unroll
      for(int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f32_32x32x2f32(x,y,acc[i],0,0,0);
  }
  float s=0; for(int i=0;i<NACC;i++) for(int r=0;r<16;r++) s+=acc[i][r];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
extern "C" void run(float* out, int grid, int block, int iters, int nacc, hipStream_t st){
  if(nacc==1) hipLaunchKernelGGL(peak<1>,dim3(grid),dim3(block),0,st,out,iters,1.f,2.f);
  if(nacc==2) hipLaunchKernelGGL(peak<2>,dim3(grid),dim3(block),0,st,out,iters,1.f,2.f);
  if(nacc==4) hipLaunchKernelGGL(peak<4>,dim3(grid),dim3(block),0,st,out,iters,1.f,2.f);
}
