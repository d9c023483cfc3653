// tools/hostbench.cpp -- host stages of one 65 536-frame list (index pass pooled / one thread, whole build, planner, packer), timed
// alone on the box.  Build: hipcc -x c++ -O2 -std=c++17 -I dcsexplorer_amd/csrc -o /tmp/hostbench tools/hostbench.cpp -L dcsexplorer_amd -ldcs_hip -Wl,-rpath,$PWD/dcsexplorer_amd
#include "dcs_common.h"
#include <chrono>
#include <cstdio>
#include <vector>
#include <cstring>
static double now(){return std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();}
int main(int argc,char**argv){
  int nStreams=256,nFrames=256; 
  std::vector<std::vector<uint8_t>> data(nStreams); std::vector<DcsStreamRef> refs(nStreams);
  for(int k=0;k<nStreams;k++){ DcsSynthParams p; memset(&p,0,sizeof p); int m=k%10; p.seed=0x94000003+k; p.format= m==0?3: m==1?4:5; p.nFrames=nFrames; p.nBands=16; p.strideFromBand=(k%7)?16:12; p.profile=0; size_t n=0; dcs_synth_stream(&p,nullptr,0,&n); data[k].resize(n); dcs_synth_stream(&p,data[k].data(),n,&n); refs[k]={data[k].data(),n,(k&1)?3:2,220,0x64,255}; }
  for(int rep=0;rep<3;rep++){
    double t0=now(); DcsBuiltStreams B; DcsStatus st=dcsBuildStreams(refs.data(),nStreams,0,B,false,false); double t1=now();
    // index alone
    std::vector<DcsFrameIndex> idx((size_t)nStreams*nFrames); std::vector<DcsStreamInfo> infos(nStreams); std::vector<uint64_t> first(nStreams); for(int k=0;k<nStreams;k++) first[k]=(uint64_t)k*nFrames;
    double t2=now(); dcs_index_streams(refs.data(),nStreams,0,idx.data(),first.data(),infos.data()); double t3=now();
    dcs_index_streams(refs.data(),nStreams,1,idx.data(),first.data(),infos.data()); double t4=now();
    std::vector<DcsSlot> slots; double t5=now(); uint32_t nc=dcsPlanChunks(B.jobs.data(),(uint32_t)B.jobs.size(),B.srcs.data(),8,slots,true); double t6=now();
    std::vector<uint8_t> out((size_t)nc*dcsPkgBytes(8)); double t7=now(); dcsBuildPackages(slots.data(),nc,8,B.srcs.data(),B.blob.data(),B.blob.size(),out.data()); double t8=now();
    // the build from records that are already there (the pipeline's device-index path)
    { DcsBuiltStreams B2; std::vector<uint64_t> off(nStreams); uint64_t o=0; for(int k=0;k<nStreams;k++){ off[k]=o; o+=(refs[k].len+3)&~size_t(3);} DcsPreIndexed pre{idx.data(), first.data(), infos.data(), off.data()};
      double a=now(); dcsBuildStreams(refs.data(),nStreams,0,B2,false,false,&pre); double b=now(); dcsBuildStreams(refs.data(),nStreams,0,B2,false,false,&pre); double c=now();
      printf("build from records: first %.2f ms, again into the same vectors %.2f ms\n", b-a, c-b); }
    printf("st=%d build(all)=%.2f ms  index pooled=%.2f  index 1thr=%.2f  plan=%.2f  pack=%.2f (alloc %.2f) chunks=%u threads=%d\n",st,t1-t0,t3-t2,t4-t3,t6-t5,t8-t7,t7-t6,nc,dcs_host_threads());
  }
}
