import torch, sys, os
sys.path.insert(0, os.getcwd())
from gpemsr_amd import ops
dev=torch.device("cuda",0)
g=torch.Generator().manual_seed(0)
n,h,w,s=80,128,128,8
ref=torch.rand(n,1,h*s,w*s,generator=g).to(dev); lr=torch.rand(n,1,h,w,generator=g).to(dev)
w1=(torch.rand(64,3,3,3,generator=g)-0.5).to(dev); b1=torch.rand(64,generator=g).to(dev)
w2=(torch.rand(64,64,3,3,generator=g)-0.5)/24; b2=torch.rand(64,generator=g).to(dev)
from gpemsr_amd.packing import pack_conv_bf16
w2b=pack_conv_bf16(w2,dev)
A=lambda t: ops.Act(t.reshape(-1),t.shape[0],t.shape[2],t.shape[3],1,1,0)
w1s=w1.sum(1).reshape(64,9).contiguous()
up=ops.bilinear(A(lr),h*s,w*s)
call=(lambda: ops.vgg_mask_bf16(A(ref),up,1,w1s,b1,w2b,b2)) if os.environ.get("GPEMSR_VGG_UPLR","1")!="0" else (lambda: ops.vgg_mask_bf16(A(ref),A(lr),s,w1s,b1,w2b,b2))
for _ in range(2): out=call()
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): out=call()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("GPEMSR_VGG_FORM","2"),os.environ.get("GPEMSR_VGG_DBG","0"),os.environ.get("GPEMSR_LIB_PATH","")[-24:],"ms",e0.elapsed_time(e1)/5)
