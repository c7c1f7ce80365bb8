#include "../cuclarabel_amd/csrc/symbolic.hpp"
using namespace hipkkt;
extern "C" int get_perm(int N,const int64_t* cp,const int64_t* ri,int ordering,int64_t* perm){
  SymbolicOptions o; o.ordering=ordering; o.nd_leaf_size=1000; Symbolic S; analyse(N,cp,ri,0,o,S);
  for(int i=0;i<N;i++) perm[i]=S.perm[i]; return 0; }
