#!/bin/bash
# bamfilt: what happens between "output closed" and the end of main (teardown inside filter()'s scope)?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/bench_bamfilt_program.py --runs 2 > /dev/null 2>&1
wd=/tmp/pjb_bamfilt
for k in 1; do
rm -f $wd/filt/f2.bam*
PJB_PROFILE_HOST=1 PORTCULLIS_PROFILE=1 portcullis_amd/host/portcullis_amd bamfilt -o $wd/filt/f2.bam -c HARD -t 16 $wd/pass.junctions.tab $wd/prep/portcullis.sorted.alignments.bam 2>&1 | tee /dev/stderr | grep "epoch\|output closed\|junctions loaded" | python -c "
import sys,re
t={}
for l in sys.stdin:
    m=re.search(r'main entered at epoch ([\d.]+)',l); 
    if m: t['in']=float(m.group(1))
    m=re.search(r'leaving main at epoch ([\d.]+)',l)
    if m: t['out']=float(m.group(1))
    m=re.search(r't=([\d.]+) s: output closed',l)
    if m: t['closed']=float(m.group(1))
print('main', round(t['out']-t['in'],3), 's; filter() closed its output at', t['closed'])"
done
