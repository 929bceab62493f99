#!/usr/bin/env python3
"""Robustness fuzz of the C++ video reader (include/lkingest.h): mutated copies of the reference's pacman.mp4 (bytes flipped in the sample
tables / parameter sets, in slice data, truncations, single bit flips in the first IDR slice) through tests/cpp/video_file_check built with
-fsanitize=address,undefined.  Every file must end in a status (0: decoded, 1: rejected), never in a sanitizer report or a signal.
usage: tools/fuzz_ingest.py [seed] [cases] [harness]      (CPU only; tests/test_h264_ingest.py runs a short slice of it)"""
import subprocess, random, sys, os
src=open('/root/reference/pacman.mp4','rb').read()
random.seed(int(sys.argv[1]) if len(sys.argv)>1 else 1)
n=int(sys.argv[2]) if len(sys.argv)>2 else 200
exe=sys.argv[3] if len(sys.argv)>3 else '/tmp/vfc_asan'
tmp=os.path.dirname(exe)
moov=src.find(b'moov')
stats={}
for it in range(n):
    b=bytearray(src)
    mode=it%4
    if mode==0:   # flip bytes inside the sample tables / avcC
        for _ in range(random.randint(1,6)):
            p=random.randint(moov, len(b)-1); b[p]=random.randint(0,255)
    elif mode==1: # flip bytes inside slice data
        for _ in range(random.randint(1,20)):
            p=random.randint(48, moov-1); b[p]=random.randint(0,255)
    elif mode==2: # truncate
        b=b[:random.randint(0,len(b)-1)]
    else:         # flip bits in the first IDR slice
        for _ in range(random.randint(1,4)):
            p=random.randint(48, 48+13000); b[p]^=1<<random.randint(0,7)
    open(tmp+'/fz.mp4','wb').write(b)
    r=subprocess.run([exe,tmp+'/fz.mp4',tmp+'/fz_l.u8',tmp+'/fz_b.u8'],capture_output=True,text=True,timeout=120)
    key=r.returncode
    stats[key]=stats.get(key,0)+1
    if r.returncode not in (0,1) or 'Sanitizer' in r.stderr or 'runtime error' in r.stderr:
        print('CRASH', it, mode, r.returncode, r.stderr[-1500:]); open(tmp+'/crash_%d.mp4'%it,'wb').write(b); sys.exit(1)
print(stats)
