import sys, os
sys.path[:0]=['/root/repo','/root/repo/diff-mst_amd','/root/repo/diff-mst_amd/standalone','/root/repo/tests']
import torch
from mst.modules import AdvancedMixConsole
from oracle import console_restated as oc
from util import FULL, rel
dev=torch.device('cuda:0')
for seed in (3,4,5):
    torch.manual_seed(seed)
    bs,T,n=1,2,131072
    tracks=0.1*torch.randn(bs,T,n); tp,fp,mp=torch.rand(bs,T,27),torch.rand(bs,25),torch.rand(bs,26); g=torch.randn(bs,2,n)
    res={}
    for multi in (False,True):
        c=AdvancedMixConsole(44100); c._multipass_eq=multi
        tr=tracks.to(dev).requires_grad_(True)
        _,mix,*_=c(tr,tp.to(dev),fp.to(dev),mp.to(dev),**FULL)
        (mix*g.to(dev)).sum().backward(); res[multi]=tr.grad.cpu()
    out={}
    for dt in (torch.float32, torch.float64):
        tr=tracks.detach().clone().to(dt).requires_grad_(True)
        _,mix,*_=oc.console_forward(tr,tp.to(dt),fp.to(dt),mp.to(dt),**FULL)
        (mix*g.to(dt)).sum().backward(); out[dt]=tr.grad
    print(f"seed {seed}: grad_tracks vs f64: MFMA-zs in-wave {rel(res[False],out[torch.float64]):.2e}, three-kernel (VALU zs) {rel(res[True],out[torch.float64]):.2e}, ref32 {rel(out[torch.float32],out[torch.float64]):.2e}; in-wave vs three-kernel {rel(res[False],res[True]):.2e}")
