import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd
from dss_amd import hip
torch.manual_seed(0)
for (b,t,heads) in [(1,64,1),(1,65,1),(1,320,1),(1,901,2)]:
    qkv = (torch.randn(b,t,3*heads*64)*1.5).half()
    out = hip.attention(qkv.cuda(), heads, 0.125).cpu().double()
    q,k,v = qkv.double().reshape(b,t,3,heads,64).permute(2,0,3,1,4)
    a = ((q@k.transpose(-1,-2))*0.125).softmax(-1)
    ref = (a@v).transpose(1,2).reshape(b,t,heads*64)
    err = (out-ref).abs()
    print((b,t,heads), 'max err', err.max().item(), 'ref max', ref.abs().max().item())
    rowerr = err.amax(-1)[0]
    bad = (rowerr>1e-2).nonzero().flatten()
    print('  bad rows', bad[:40].tolist(), 'count', len(bad))
    if len(bad):
        r = bad[0].item()
        ratio = (out[0,r]/ref[0,r])
        print('  ratio out/ref row', r, ratio[:8].tolist())
