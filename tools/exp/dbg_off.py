import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from nunet_amd import NutlsOffline
clip = np.load("tests/golden/clip_4s.npz")["mags_in"]
def rms(a,b): return float(np.sqrt(np.mean((np.asarray(a,np.float64)-np.asarray(b,np.float64))**2)))
for (U,T,sizes,C) in ((3,24,[24,7,24,1,17],2),(2,600,[600,333],0),(3,300,[300,299],3),(2,40,[33,40],4)):
    x = np.stack([np.concatenate([clip, clip, clip, clip])[40*u:40*u+sum(sizes)] for u in range(U)])
    off = NutlsOffline(max_frames=T, utterances=U, pipeline=C)
    outs=[];t=0
    for n in sizes:
        outs.append(off.process(x[:,t:t+n])); t+=n
    got=np.concatenate(outs,axis=1)
    errs=[]
    for u in range(U):
        one = NutlsOffline(max_frames=T)
        w = one.process(x[u]); one.close()
        errs.append([round(rms(got[u,i],w[i]),7) for i in range(sum(sizes))])
    off.close()
    print(U,T,sizes,"chunks",C, "max per-frame rms per utterance:", [max(e) for e in errs], "first bad frame:", [next((i for i,v in enumerate(e) if v>1e-5), None) for e in errs])
