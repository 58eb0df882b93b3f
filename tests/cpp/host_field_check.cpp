// CPU check of the host-side field / group arithmetic that finishes every MSM (gemini_amd/csrc/host_field.hpp): the BMI2 + ADX
// Montgomery product of host_fq_adx.hpp against the portable loop on random and boundary operands, the group law through both,
// and the timing of the window Horner's doubling chain.  Built and run by tests/test_host_field_cpu.py (g++ and, when present,
// the ROCm clang++); GM_HOST_ADX=0 forces the portable path.
#include "host_field.hpp"
#include <chrono>
#include <cstdio>
#include <random>
using namespace gmh;
int main(int argc, char** argv){
  const int iters = argc > 1 ? atoi(argv[1]) : 300000;
  std::mt19937_64 rng(7);
  Fq a,b; for(int i=0;i<6;i++){a.l[i]=rng();b.l[i]=rng();} a.l[5]&=0x0fffffffffffffffull; b.l[5]&=0x0fffffffffffffffull;
  a = a.mul_generic(Fq::one()); b=b.mul_generic(Fq::one());
#ifdef GM_HAVE_FQ_ADX
  printf("adx usable: %d\n", (int)fq_adx_usable());
#else
  printf("adx not compiled in\n");
#endif
  for(int it=0;it<iters;it++){
    Fq c=a.mul_generic(b), d=a*b;
    if(!(c==d)){printf("MISMATCH at %d\n",it);return 1;}
    a=b; b=c; if(it%5==0) b=b+a; if(it%11==0) a=a-b;
  }
  Fq m1; memcpy(m1.l,FqP::MOD,48); m1.l[0]-=1; Fq z=Fq::zero(), o=Fq::one();
  Fq cases[4]={m1,z,o,a};
  for(auto&x:cases)for(auto&y:cases){ if(!(x.mul_generic(y)==x*y)){printf("EDGE MISMATCH\n");return 1;}}
  // a group-law check through both paths: 2P + P == 3P via dbl/add on the generator
  u64 gx[6]={0x5cb38790fd530c16ull,0x7817fc679976fff5ull,0x154f95c7143ba1c1ull,0xf0ae6acdf3d0e747ull,0xedce6ecc21dbf440ull,0x120177419e0bfb75ull};
  u64 gy[6]={0xbaac93d50ce72271ull,0x8c22631a7918fd8eull,0xdd595f13570725ceull,0x51ac582950405194ull,0x0e1c8c3fad0059c0ull,0x0bbc3efc5008a26aull};
  G1 g; memcpy(g.x.l,gx,48); memcpy(g.y.l,gy,48); g.z=Fq::one();
  G1 p3a = g.dbl().add(g).normalized(), p3b = g.add(g).add(g).normalized();
  if(!(p3a.x==p3b.x && p3a.y==p3b.y)){printf("GROUP MISMATCH\n");return 1;}
  Fr x=Fr::one(); Fr y=x*x; if(!(y==x)){printf("FR MISMATCH\n");return 1;}
  printf("ok\n");
  G1 r=g;
  auto t0=std::chrono::steady_clock::now();
  for(int it=0;it<100;it++){ for(int i=0;i<256;i++) r=r.dbl(); r=r.add(g);} 
  auto t1=std::chrono::steady_clock::now();
  printf("256 dbl + add: %.1f us  (z0=%llx)\n", std::chrono::duration<double,std::micro>(t1-t0).count()/100,(unsigned long long)r.z.l[0]);
  t0=std::chrono::steady_clock::now();
  G1 n=r.normalized();
  t1=std::chrono::steady_clock::now();
  printf("normalise: %.1f us (%llx)\n", std::chrono::duration<double,std::micro>(t1-t0).count(),(unsigned long long)n.x.l[0]);
  return 0;
}
