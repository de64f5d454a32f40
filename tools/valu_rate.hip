// VALU issue-rate microbenchmark: N dependent-free chains of one instruction kind per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("hip error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
template<int KIND> __global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b){
  float x0=threadIdx.x*1e-3f+a,x1=x0+1,x2=x0+2,x3=x0+3,x4=x0+4,x5=x0+5,x6=x0+6,x7=x0+7;
  unsigned u0=threadIdx.x,u1=u0+1,u2=u0+2,u3=u0+3,u4=u0+4,u5=u0+5,u6=u0+6,u7=u0+7;
  for(int i=0;i<iters;i++){
    if(KIND==0){ // fma
      x0=__builtin_fmaf(x0,a,b);x1=__builtin_fmaf(x1,a,b);x2=__builtin_fmaf(x2,a,b);x3=__builtin_fmaf(x3,a,b);
      x4=__builtin_fmaf(x4,a,b);x5=__builtin_fmaf(x5,a,b);x6=__builtin_fmaf(x6,a,b);x7=__builtin_fmaf(x7,a,b);
    } else if(KIND==1){ // rcp
      x0=__builtin_amdgcn_rcpf(x0);x1=__builtin_amdgcn_rcpf(x1);x2=__builtin_amdgcn_rcpf(x2);x3=__builtin_amdgcn_rcpf(x3);
      x4=__builtin_amdgcn_rcpf(x4);x5=__builtin_amdgcn_rcpf(x5);x6=__builtin_amdgcn_rcpf(x6);x7=__builtin_amdgcn_rcpf(x7);
    } else if(KIND==2){ // cndmask-ish: select
      x0=x0>b?x0*a:x1;x1=x1>b?x1*a:x2;x2=x2>b?x2*a:x3;x3=x3>b?x3*a:x4;x4=x4>b?x4*a:x5;x5=x5>b?x5*a:x6;x6=x6>b?x6*a:x7;x7=x7>b?x7*a:x0;
    } else if(KIND==3){ // mul_lo_u32
      u0*=u1|1;u1*=u2|1;u2*=u3|1;u3*=u4|1;u4*=u5|1;u5*=u6|1;u6*=u7|1;u7*=u0|1;
    } else if(KIND==4){ // add_u32 + xor + rot (ARX)
      u0+=u1;u1^=u0;u1=(u1<<7)|(u1>>25);u2+=u3;u3^=u2;u3=(u3<<9)|(u3>>23);u4+=u5;u5^=u4;u5=(u5<<13)|(u5>>19);u6+=u7;u7^=u6;u7=(u7<<11)|(u7>>21);
    } else if(KIND==6){ // packed fma: 2 FMAs per instruction
      typedef float v2 __attribute__((ext_vector_type(2)));
      v2 a2={a,a}, b2={b,b};
      v2 p0={x0,x1},p1={x2,x3},p2={x4,x5},p3={x6,x7};
      asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   : "+v"(p0),"+v"(p1),"+v"(p2),"+v"(p3) : "v"(a2),"v"(b2));
      x0=p0.x;x1=p0.y;x2=p1.x;x3=p1.y;x4=p2.x;x5=p2.y;x6=p3.x;x7=p3.y;
    } else if(KIND==5){ // mul f32
      x0*=a;x1*=a;x2*=a;x3*=a;x4*=a;x5*=a;x6*=a;x7*=a;
    }
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=x0+x1+x2+x3+x4+x5+x6+x7+(float)(u0^u1^u2^u3^u4^u5^u6^u7);
}
template<int KIND> int run(const char* name,int opsPerIter,int wavesPerSimd){
  int blocks=256*wavesPerSimd; // 256 CUs x (4 waves per block -> 1 wave per SIMD per block)
  float* out; CHECK(hipMalloc(&out,blocks*256*4));
  int iters=20000;
  hipEvent_t e0,e1; CHECK(hipEventCreate(&e0));CHECK(hipEventCreate(&e1));
  k<KIND><<<blocks,256>>>(out,100,1.0001f,0.5f); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0)); k<KIND><<<blocks,256>>>(out,iters,1.0001f,0.5f); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms,e0,e1));
  double instr=(double)blocks*4*iters*opsPerIter; // wave-instructions
  double perSimdPerUs=instr/1024.0/(ms*1e3);
  printf("%-10s waves/SIMD %d: %.1f ms, %.1f wave-instr/us/SIMD -> %.2f cycles/instr at 2.4 GHz\n",name,wavesPerSimd,ms,perSimdPerUs,2400.0/perSimdPerUs);
  CHECK(hipFree(out)); return 0;
}
int main(){
  for(int w: {1,2,4,8}){ run<0>("fma",8,w); }
  for(int w: {1,2,4,8}){ run<6>("pk_fma(x2)",8,w); }
  for(int w: {1,4}){ run<5>("mul",8,w); run<1>("rcp",8,w); run<2>("cmp+sel+mul",24,w); run<3>("mul_lo_u32",16,w); run<4>("arx",16,w); }
}
