// scratch: run symbolic analysis on a triu CSC dumped to binary, print stats
#include "../cuclarabel_amd/csrc/symbolic.hpp"
#include <cstdio>
#include <chrono>
#include <algorithm>
using namespace hipkkt;
extern "C" int sym_stats(int N, const int64_t* cp, const int64_t* ri, int ordering, int leaf, double* out, int c0,int c1,int c2,double z1,double z2,double z3) {
    SymbolicOptions o; o.ordering = ordering; o.nd_leaf_size = leaf;
    o.relax_cols[0]=c0;o.relax_cols[1]=c1;o.relax_cols[2]=c2;o.relax_zeros[1]=z1;o.relax_zeros[2]=z2;o.relax_zeros[3]=z3;
    Symbolic S;
    auto t0 = std::chrono::steady_clock::now();
    analyse(N, cp, ri, 0, o, S);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
    printf("N=%d nnzK=%ld nnzL_struct=%ld nnzL=%ld flops=%.3g etree_h=%d nsuper=%d nlevels=%zu maxfront=%d front_store=%.1fMB upd_store=%.1fMB time=%.2fs\n",
        S.N,(long)S.nnzK,(long)S.nnzL_struct,(long)S.nnzL,S.flops,S.etree_height,S.nsuper,S.levels.size(),S.max_front,S.front_store*8e-6,S.update_store*8e-6,dt);
    // histogram of nc and f
    int hb[8]={1,2,4,8,16,32,64,1<<30}; long hc[8]={0}; long hf[8]={0};
    int fb[8]={8,16,32,64,128,256,512,1<<30};
    for(int s=0;s<S.nsuper;s++){int nc=S.sn_start[s+1]-S.sn_start[s]; int f=nc+(int)(S.rowptr[s+1]-S.rowptr[s]);
      for(int b=0;b<8;b++) if(nc<=hb[b]){hc[b]++;break;}
      for(int b=0;b<8;b++) if(f<=fb[b]){hf[b]++;break;}}
    printf("nc hist (<=1,2,4,8,16,32,64,inf):"); for(int b=0;b<8;b++) printf(" %ld",hc[b]); printf("\n");
    printf("f  hist (<=8,16,32,64,128,256,512,inf):"); for(int b=0;b<8;b++) printf(" %ld",hf[b]); printf("\n");
    // per-level: count, max f, sum flops
    size_t nl=S.levels.size();
    for(size_t l=0;l<nl;l++){  int cnt=S.levels[l].end-S.levels[l].begin; int mf=0,mnc=0; double fl=0;
      for(int t=S.levels[l].begin;t<S.levels[l].end;t++){int s=S.level_sn[t]; int nc=S.sn_start[s+1]-S.sn_start[s]; int nb=(int)(S.rowptr[s+1]-S.rowptr[s]); mf=std::max(mf,nc+nb); mnc=std::max(mnc,nc); fl+= (double)nc*(nc+nb)*(nc+nb);} 
      printf("  level %zu: %d fronts, max f=%d max nc=%d ~flops %.3g\n",l,cnt,mf,mnc,fl);}
    out[0]=S.nnzL; out[1]=S.flops; out[2]=S.nsuper; out[3]=S.levels.size();
    return 0;
}
