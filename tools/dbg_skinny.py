import sys; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
from layoutdetr_amd.hip import core
dev=torch.device('cuda:0')
for (M,N,K) in [(4096,192,64),(4096,256,128),(4096,256,64),(4096,512,64),(4096,160,64)]:
    torch.manual_seed(1)
    A=torch.randn(M,K); W=torch.randn(N,K)
    c=torch.full((M,N),-7.0,device=dev)
    core.gemm(A.to(dev),W.to(dev),0,0,M,N,K,out=c)
    ref=A@W.t()
    d=(c.cpu()-ref).abs()
    bad=(d>1e-3)
    print(M,N,K,'bad frac',bad.float().mean().item(),'untouched',(c.cpu()==-7).float().mean().item())
    if bad.any():
        print(' per-32col bad frac', [round(bad[:,i:i+32].float().mean().item(),2) for i in range(0,N,32)])
        print(' per-32row bad frac (first 8)', [round(bad[i:i+32].float().mean().item(),2) for i in range(0,256,32)])
        print(' c[0,:8]', c[0,:8].tolist()); print(' ref[0,:8]', ref[0,:8].tolist())
        # is c a permutation / other element of ref?
        v=c[0,0].item(); idx=((ref-v).abs()<1e-4).nonzero()
        print(' c[0,0] found in ref at', idx[:5].tolist())
