// The fast row formatters (Junction::appendTabRow / appendBedRow) must produce exactly the bytes of the
// iostream formatters that mirror the reference (operator<<, outputBED) on random junction rows.
#include <portcullis/junction_system.hpp>
#include "portcullis_amd.h"
#include <sstream>
#include <cstring>
#include <random>
#include <iostream>
using namespace portcullis;
int main(){
  auto refs = std::make_shared<bam::RefSeqPtrList>();
  refs->push_back(std::make_shared<bam::RefSeq>(0,"chrA",1000000));
  std::mt19937_64 rng(5);
  int bad=0;
  for(int it=0; it<20000; it++){
    pjb_junction_row r; memset(&r,0,sizeof r);
    r.refid=0; r.start=1000+rng()%500000; r.end=r.start+rng()%1000; r.left=r.start-(rng()%100); r.right=r.end+rng()%100;
    r.read_strand=rng()%3; r.ss_strand=rng()%3; r.cons_strand=rng()%3; r.canonical=rng()%3;
    r.da1[0]='G'; r.da1[1]=(it%50==0)?0:'T'; r.da2[0]='A'; r.da2[1]='G'; r.suspicious=rng()%2;
    r.nb_raw=1+rng()%(it%3==0?100:4000000); r.nb_dist=rng()%r.nb_raw+1; r.nb_ms=rng()%r.nb_raw; r.nb_um=rng()%(r.nb_raw+1); r.nb_bpp=rng()%100; r.nb_ppp=rng()%100; r.nb_rel=rng()%(r.nb_raw+1);
    r.r1pos=rng()%1000; r.r1neg=rng()%1000; r.r2pos=rng()%10; r.r2neg=rng()%5;
    r.max_min_anc=rng()%100; r.maxmmes=rng()%100; r.hamming5p=rng()%11; r.hamming3p=rng()%11; r.nb_up_juncs=rng()%3; r.nb_down_juncs=rng()%3;
    for(int k=0;k<20;k++) r.jad[k]=rng()%r.nb_raw;
    r.sum_mismatches=rng()%(it%2?100000000ull:100); r.entropy=(double)(rng()%1000000)/ (double)(1+rng()%1000000) * (it%7==0?1e-5:1.0);
    auto j = Junction::fromRow(r,*refs);
    j->setId(it); j->setMeanReadLength(rng()%300); j->setDistanceToNextUpstreamJunction(it%5==0?(uint32_t)-1:rng()%100000); j->setUniqueJunction(it%2); j->setPrimaryJunction(it%3==0); j->setPotentialFalsePositive(it%11==0);
    std::ostringstream a; a << *j; std::string b; j->appendTabRow(b);
    std::ostringstream c; j->outputBED(c,"portcullis",false); std::string d; j->appendBedRow(d,"portcullis",false);
    if(a.str()!=b || c.str()!=d){ if(bad<3) std::cerr<<a.str()<<"\n"<<b<<"\n"<<c.str()<<d<<"\n"; bad++; }
  }
  std::cout << "mismatches: " << bad << std::endl; return bad!=0;
}
