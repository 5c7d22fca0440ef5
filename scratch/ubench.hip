// micro-benchmarks: issue rate of v_qsad_pk_u16_u8 / v_sad_u8 / v_alignbyte on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
template<int MODE> __global__ void k(uint64_t* out, int iters, uint32_t seed){
  uint64_t a0=seed+threadIdx.x, a1=a0*3, a2=a0*5, a3=a0*7, a4=a0*11, a5=a0*13, a6=a0*17, a7=a0*19;
  uint64_t w = 0x0123456789abcdefull * (threadIdx.x+1); uint32_t f = seed*2654435761u;
  uint32_t s0=a0,s1=a1,s2=a2,s3=a3,s4=a4,s5=a5,s6=a6,s7=a7;
  long long t0 = clock64();
  for(int i=0;i<iters;i++){
    if(MODE==0){
      a0=__builtin_amdgcn_qsad_pk_u16_u8(w,f,a0); a1=__builtin_amdgcn_qsad_pk_u16_u8(w,f,a1);
      a2=__builtin_amdgcn_qsad_pk_u16_u8(w,f,a2); a3=__builtin_amdgcn_qsad_pk_u16_u8(w,f,a3);
      a4=__builtin_amdgcn_qsad_pk_u16_u8(w,f,a4); a5=__builtin_amdgcn_qsad_pk_u16_u8(w,f,a5);
      a6=__builtin_amdgcn_qsad_pk_u16_u8(w,f,a6); a7=__builtin_amdgcn_qsad_pk_u16_u8(w,f,a7);
    } else if(MODE==1){
      s0=__builtin_amdgcn_sad_u8((uint32_t)w,f,s0); s1=__builtin_amdgcn_sad_u8((uint32_t)w,f,s1);
      s2=__builtin_amdgcn_sad_u8((uint32_t)w,f,s2); s3=__builtin_amdgcn_sad_u8((uint32_t)w,f,s3);
      s4=__builtin_amdgcn_sad_u8((uint32_t)w,f,s4); s5=__builtin_amdgcn_sad_u8((uint32_t)w,f,s5);
      s6=__builtin_amdgcn_sad_u8((uint32_t)w,f,s6); s7=__builtin_amdgcn_sad_u8((uint32_t)w,f,s7);
    } else if(MODE==2){
      s0=__builtin_amdgcn_alignbyte(s0,f,1); s1=__builtin_amdgcn_alignbyte(s1,f,2);
      s2=__builtin_amdgcn_alignbyte(s2,f,3); s3=__builtin_amdgcn_alignbyte(s3,f,1);
      s4=__builtin_amdgcn_alignbyte(s4,f,2); s5=__builtin_amdgcn_alignbyte(s5,f,3);
      s6=__builtin_amdgcn_alignbyte(s6,f,1); s7=__builtin_amdgcn_alignbyte(s7,f,2);
    } else {
      a0=__builtin_amdgcn_mqsad_pk_u16_u8(w,f,a0); a1=__builtin_amdgcn_mqsad_pk_u16_u8(w,f,a1);
      a2=__builtin_amdgcn_mqsad_pk_u16_u8(w,f,a2); a3=__builtin_amdgcn_mqsad_pk_u16_u8(w,f,a3);
      a4=__builtin_amdgcn_mqsad_pk_u16_u8(w,f,a4); a5=__builtin_amdgcn_mqsad_pk_u16_u8(w,f,a5);
      a6=__builtin_amdgcn_mqsad_pk_u16_u8(w,f,a6); a7=__builtin_amdgcn_mqsad_pk_u16_u8(w,f,a7);
    }
  }
  long long t1 = clock64();
  uint64_t r = a0^a1^a2^a3^a4^a5^a6^a7^s0^s1^s2^s3^s4^s5^s6^s7;
  out[blockIdx.x*blockDim.x+threadIdx.x] = r;
  if(threadIdx.x==0 && blockIdx.x==0) out[1<<20] = (uint64_t)(t1-t0);
}
template<int MODE> int run(const char* name, uint64_t* d, int threads){
  int iters=20000; hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256*4), dim3(threads), 0, 0, d, 100, 1u); CHK(hipDeviceSynchronize());
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256*4), dim3(threads), 0, 0, d, iters, 7u); hipEventRecord(e1); CHK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms,e0,e1); uint64_t cyc; hipMemcpy(&cyc, d+(1<<20), 8, hipMemcpyDeviceToHost);
  double inst = 8.0*iters; printf("%-10s threads/blk %4d: %.1f clk64 ticks/inst (wave0)  wall %.3f ms\n", name, threads, (double)cyc/inst, ms);
  return 0;
}
int main(){ uint64_t* d; CHK(hipMalloc(&d, ((1<<20)+16)*8));
  for(int th : {64,256,512}){ run<0>("qsad",d,th); run<1>("sad_u8",d,th); run<2>("alignbyte",d,th); run<3>("mqsad",d,th);} return 0; }
