#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static uint64_t s[2]={0x9E3779B97F4A7C15ull,0xD1B54A32D192ED03ull};
static inline uint64_t nxt(void){uint64_t a=s[0],b=s[1];s[0]=b;a^=a<<23;s[1]=a^b^(a>>17)^(b>>26);return s[1]+b;}
static inline double mk(uint64_t mant,int e){uint64_t bits=((uint64_t)(e+1023)<<52)|(mant&((1ull<<52)-1));double d;memcpy(&d,&bits,8);return d;}
int main(){
  long bad=0,n=0;
  for(long it=0;it<400000000L;++it){
    uint64_t r1=nxt(),r2=nxt();
    double b; 
    int mode=it&7;
    if(mode==0) b=mk(~0ull-(r2&0xff),(int)(r1>>60)-8);       // mantissa near all ones
    else if(mode==1) b=mk(r2&0xff,(int)(r1>>60)-8);           // near power of two
    else b=mk(r2,(int)((r1>>56)&31)-16);
    double a=mk(r1,(int)((r2>>56)&63)-32); if(r1&(1ull<<55)) a=-a;
    double rb=1.0/b;
    double q0=a*rb, r=fma(-q0,b,a), q=fma(r,rb,q0);
    double t=a/b;
    if(q!=t){ if(bad<10) printf("a=%a b=%a q=%a t=%a\n",a,b,q,t); ++bad;}
    ++n;
  }
  printf("n=%ld bad=%ld\n",n,bad);return 0;}
