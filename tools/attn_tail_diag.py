"""Which of the 1025th QUERY (a ragged last block) and the 1025th KEY (a 17th tile for every workgroup) costs the forward what, at B = 64, H = 12 (same strides in all four cases)."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from bridgeqa_amd import _ext
def med(f, n=10):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts)//2]
B,H=64,12
qkv=torch.randn(B,1025,3,H,64,device='cuda').to(torch.bfloat16)
q,k,v=qkv[:,:,0],qkv[:,:,1],qkv[:,:,2]
for lq,lk in ((1024,1024),(1025,1024),(1024,1025),(1025,1025),(1024,1024),(1025,1025),(1025,1024),(1024,1025)):
    t=med(lambda: _ext.attn_fwd(q[:,:lq],k[:,:lk],v[:,:lk],0.125))
    print("Lq=%d Lk=%d  %.1f us"%(lq,lk,t*1e3))
