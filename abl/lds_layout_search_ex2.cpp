#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstring>
// exchange-2 layout search.  element (a, k2, d): A = a*S2 + k2*Q + 2*((d>>1) ^ x(a,k2)) + (d&1), x = GF(2)-linear 2-bit function of (a0..a2,k0..k3)
static int R128[4][16]={{0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27},{4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31},
 {32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59},{36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63}};
static inline int klo(int l){return l?128-l:64;}
int S2,Q,M0,M1;
static inline int xf(int a,int k2){int v=a|(k2<<3);return (__builtin_parity(v&M0))|(__builtin_parity(v&M1)<<1);}
static inline int A(int a,int k2,int d){return a*S2+k2*Q+2*(((d>>1)^xf(a,k2)))+(d&1);}
int ways(const int* addrs,int n,int ndw_complex,int nbanks_complex){ // addrs in complex units; each access covers ndw_complex consecutive complex; banks at complex granularity
  int cnt[64];int seen[64][8];memset(cnt,0,sizeof(cnt));
  int w=0;
  for(int i=0;i<n;i++)for(int k=0;k<ndw_complex;k++){int ad=addrs[i]+k;int b=ad%nbanks_complex;bool f=false;for(int j=0;j<cnt[b];j++)if(seen[b][j]==ad)f=true;if(!f){if(cnt[b]<8)seen[b][cnt[b]]=ad;cnt[b]++;if(cnt[b]>w)w=cnt[b];}}
  return w;
}
int main(){
  int best=1e9;
  for(int s2=0;s2<32;s2+=2)for(Q=8;Q<=16;Q+=2)for(M0=0;M0<128;M0++)for(M1=0;M1<128;M1++){
    S2=160+s2; if(S2<16*Q) S2+=32; if(S2<16*Q) continue;
    int cost=0;
    int ad[64];
    // side A b128: lane (a3,g3), instr (h,c)
    int r128=0,w128=0;
    for(int h=0;h<2;h++)for(int c=0;c<4;c++){
      int tot=0;
      for(int g=0;g<4;g++){for(int i=0;i<16;i++){int l=R128[g][i];ad[i]=A(l&7,(l>>3)+8*h,2*c);} tot+=ways(ad,16,2,32);}
      r128+=tot-4;
      tot=0;
      for(int g=0;g<8;g++){for(int i=0;i<8;i++){int l=8*g+i;ad[i]=A(l&7,(l>>3)+8*h,2*c);} tot+=ways(ad,8,2,16);}
      w128+=std::max(0,tot-13);
    }
    int r64=0,w64=0;
    for(int a=0;a<8;a++)for(int which=0;which<2;which++){
      int tot=0;
      for(int g=0;g<2;g++){for(int i=0;i<32;i++){int l=32*g+i;int ii=which?klo(l):l;ad[i]=A(a,ii&15,ii>>4);} tot+=ways(ad,32,1,32);}
      r64+=tot-2;
      tot=0;
      for(int g=0;g<4;g++){for(int i=0;i<16;i++){int l=16*g+i;int ii=which?klo(l):l;ad[i]=A(a,ii&15,ii>>4);} tot+=ways(ad,16,1,16);}
      w64+=std::max(0,tot-6);
    }
    cost=r128+w128+r64+w64;
    if(cost<best){best=cost;printf("cost %d (r128 %d w128 %d r64 %d w64 %d) S2%%32=%d Q=%d M0=%d M1=%d\n",cost,r128,w128,r64,w64,s2,Q,M0,M1);}
  }
}
