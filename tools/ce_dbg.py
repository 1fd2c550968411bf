import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, torch.nn.functional as F
from pinthememory_amd.hip import kernels as K
def rnd(*s, seed=0): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
nhwc=lambda t: t.permute(0,2,3,1).contiguous().cuda()
nchw=lambda t: t.permute(0,3,1,2).cpu()
for hw,HW,temp,C in [((3,300),(7,611),1.0,19),((2,130),(5,2100),1.0,19),((4,4),(4,700),1.0,5),((12,12),(48,48),1.0,19),((6,150),(14,305),1.0,19),((6,300),(14,1211),1.0,19)]:
    n=2
    lg=rnd(n,C,*hw,seed=1)*3
    g=torch.Generator().manual_seed(2)
    lab=torch.randint(0,C,(n,*HW),generator=g); lab[torch.rand(n,*HW,generator=g)<0.1]=255; lab[:,:2]=255
    for dt in (torch.float32, torch.float64):
        lr=lg.to(dt).clone().requires_grad_(True)
        loss=F.cross_entropy(F.interpolate(lr/temp,size=HW,mode='bilinear',align_corners=True),lab,ignore_index=255)
        (loss*1.7).backward()
        if dt==torch.float32: g32=lr.grad.clone()
        else: g64=lr.grad.clone()
    lgg=K.new((n,hw[0],hw[1],C),torch.zeros(1,device='cuda'),pitch_pad=True); lgg.copy_(nhwc(lg))
    out_f,field=K.upsample_ce_fwd_field(lgg,lab.cuda(),1.0/temp)
    dlf=nchw(K.upsample_ce_bwd_field(lgg,HW,out_f,field,torch.tensor([1.7],device='cuda'),1.0/temp)).double()
    r=lambda a,b:(a-b).abs().max().item()/b.abs().max().item()
    print(hw,HW,'hip vs f64 %.2e  torch32 vs f64 %.2e  hip vs torch32 %.2e'%(r(dlf,g64),r(g32.double(),g64),r(dlf,g32.double())), 'max|g| %.2e'%g64.abs().max().item())
