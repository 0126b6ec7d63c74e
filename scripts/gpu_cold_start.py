import time, sys, os
t0=time.perf_counter()
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from __graft_entry__ import load_package
pkg = load_package(); L = pkg.lib()
torch.cuda.set_device(0); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
t1=time.perf_counter()
import orc
img = torch.randint(0,256,(4,270,480,3),dtype=torch.uint8,device='cuda')
res=[]
for mode,rm in ((1,0),(5,2),(0,0),(2,0)):
    ta=time.perf_counter()
    fr=[pkg.frame_setup(img[i].data_ptr(),480,270,80,24,rm,False,False,False) for i in range(4)]
    plan=pkg.Plan(mode, orc.PALETTE_STANDARD, fr)
    out=torch.zeros(4*plan.stride,dtype=torch.uint8,device='cuda'); ln=torch.zeros(4,dtype=torch.int32,device='cuda')
    plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    tb=time.perf_counter()
    plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    tc=time.perf_counter()
    res.append((mode, round((tb-ta)*1e3,2), round((tc-tb)*1e3,3)))
print(os.environ.get('ASCIICHAT_HIP_LIB','HEAD').split('/')[-1], 'first plan+render ms / second render ms per mode:', res)
