// Host check of the signed-lane field arithmetic of lurk_amd/csrc/babybear.h (mul_s, inv, ef_mul, ef_scale, ef_inv) against 128-bit
// integer arithmetic: random operands, edge values, every sign pattern of extreme (+-p/2) operands.  Built and run by tests/test_field_host.py.
#include "babybear.h"  // -I lurk_amd/csrc
#include <cstdio>
#include <cstdlib>
#include <random>
typedef unsigned __int128 u128;
static const uint64_t P = bb::P;
uint64_t mm(uint64_t a, uint64_t b){ return (u128)a*b % P; }
uint64_t pw(uint64_t a, uint64_t e){ uint64_t r=1; while(e){ if(e&1) r=mm(r,a); a=mm(a,a); e>>=1;} return r; }
uint64_t fromm(uint32_t x){ return mm(x, pw((uint64_t)bb::R1, P-2)); }
uint32_t tom(uint64_t x){ return (uint32_t)mm(x, bb::R1); }
struct E { uint64_t c[4]; };
E emul(E a, E b){ E r{{0,0,0,0}}; for(int i=0;i<4;i++)for(int j=0;j<4;j++){ uint64_t t=mm(a.c[i],b.c[j]); if(i+j>=4) t=mm(t,11); r.c[(i+j)&3]=(r.c[(i+j)&3]+t)%P;} return r; }
int main(){
  std::mt19937_64 g(1); int bad=0;
  auto rnd=[&](int k)->uint32_t{ uint32_t edge[]={0,1,P-1,(uint32_t)(P-1)/2,(uint32_t)(P-1)/2+1,(uint32_t)(P-1)/2-1,2,(uint32_t)P-2}; if(k%5==0) return edge[g()%8]; return (uint32_t)(g()%P); };
  for(int it=0; it<400000; it++){
    bb::ef a{{rnd(it),rnd(it+1),rnd(it+2),rnd(it+3)}}, b{{rnd(it+4),rnd(it),rnd(it+1),rnd(it+7)}};
    E ea{{fromm(a.c[0]),fromm(a.c[1]),fromm(a.c[2]),fromm(a.c[3])}}, eb{{fromm(b.c[0]),fromm(b.c[1]),fromm(b.c[2]),fromm(b.c[3])}};
    if (it < 150000) {
    E er=emul(ea,eb); bb::ef r=bb::ef_mul(a,b);
    for(int k=0;k<4;k++) if(r.c[k]!=tom(er.c[k])){ if(bad++<5) printf("ef_mul mismatch it %d k %d\n",it,k);} 
    bb::ef s=bb::ef_scale(a,b.c[0]); for(int k=0;k<4;k++) if(s.c[k]!=tom(mm(ea.c[k],eb.c[0]))){ if(bad++<5) printf("scale mismatch\n"); }
    uint32_t iv=bb::inv(a.c[0]); uint64_t want = ea.c[0]? pw(ea.c[0],P-2):0; if(iv!=tom(want)){ if(bad++<5) printf("inv mismatch %u\n",a.c[0]); }
    bb::ef ai=bb::ef_inv(a); bb::ef one=bb::ef_mul(a,ai); bool z=bb::ef_is_zero(a);
    if(!z && !(one.c[0]==bb::R1&&one.c[1]==0&&one.c[2]==0&&one.c[3]==0)){ if(bad++<5) printf("ef_inv mismatch it %d\n",it);} 
    if(z && !bb::ef_is_zero(ai)) { if(bad++<5) printf("ef_inv(0)\n"); }
    for(int k=0;k<4;k++) if(ai.c[k]>=P) {bad++; printf("noncanonical\n");}
    } else {
      // range stress: worst-case centred magnitudes
      bb::ef r=bb::ef_mul(a,b); for(int k=0;k<4;k++) if(r.c[k]>=P) { if(bad++<5) printf("range\n"); }
    }
  }
  // extreme operands: all +-(p-1)/2
  uint32_t hs[2]={(uint32_t)(P-1)/2,(uint32_t)(P-1)/2+1};
  for(int m=0;m<256;m++){ bb::ef a,b; for(int k=0;k<4;k++){a.c[k]=hs[(m>>k)&1]; b.c[k]=hs[(m>>(k+4))&1];}
    E ea{{fromm(a.c[0]),fromm(a.c[1]),fromm(a.c[2]),fromm(a.c[3])}}, eb{{fromm(b.c[0]),fromm(b.c[1]),fromm(b.c[2]),fromm(b.c[3])}};
    E er=emul(ea,eb); bb::ef r=bb::ef_mul(a,b); for(int k=0;k<4;k++) if(r.c[k]!=tom(er.c[k])){ if(bad++<5) printf("extreme mismatch\n"); }
    bb::ef ai=bb::ef_inv(a); bb::ef one=bb::ef_mul(a,ai); if(!(one.c[0]==bb::R1&&one.c[1]==0&&one.c[2]==0&&one.c[3]==0)) { if(bad++<5) printf("extreme inv\n"); }
  }
  printf("bad=%d\n",bad); return bad!=0;
}
