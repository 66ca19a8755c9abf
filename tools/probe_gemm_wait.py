import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from tspo_amd import _lib
dev="cuda"; M=257*1024
g=torch.Generator(device=dev).manual_seed(0)
for name,N,K,act in [("qkv",3072,1024,0),("fc1",4096,1024,1),("fc2like",1024,4096,0)]:
    A=torch.randn(M,K,generator=g,device=dev).to(torch.bfloat16); W=(torch.randn(N,K,generator=g,device=dev)*0.03).to(torch.bfloat16)
    bias=torch.randn(N,generator=g,device=dev); out=torch.empty(M,N,dtype=torch.bfloat16,device=dev)
    dbg=torch.zeros(256*8*4,device=dev)
    for _ in range(2):
        rc=_lib.lib().tspo_gemm_bf16(C.c_void_p(A.data_ptr()),C.c_void_p(W.data_ptr()),C.c_void_p(bias.data_ptr()),C.c_void_p(dbg.data_ptr()),C.c_void_p(out.data_ptr()),1,M,N,K,act|(69<<8),None)
    torch.cuda.synchronize(); assert rc==0, _lib.lib().tspo_last_error()
    d=dbg.view(256,8,4).cpu()
    vm,bar,tot,its=d[...,0],d[...,1],d[...,2],d[...,3]
    print(f"{name}: per K-step cycles: total {float((tot/its).mean()):.0f}  vmcnt-wait {float((vm/its).mean()):.0f}  barrier-wait {float((bar/its).mean()):.0f}   (wave0 vm {float((vm[:,0]/its[:,0]).mean()):.0f}, others {float((vm[:,1:]/its[:,1:]).mean()):.0f})")
