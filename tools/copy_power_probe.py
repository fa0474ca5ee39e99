import torch, time, subprocess, threading, json
x = torch.empty(1<<30, dtype=torch.uint8, device='cuda'); y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
out=[]
def samp():
    time.sleep(1.0)
    for i in range(4):
        r=subprocess.run(['rocm-smi','--showclocks','--showpower','--json'],capture_output=True,text=True).stdout
        out.append(r.strip()); time.sleep(0.4)
t=threading.Thread(target=samp); t.start()
t0=time.time(); n=0
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
while time.time()-t0 < 4.0:
    for _ in range(20): y.copy_(x)
    n+=20
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)
t.join()
print('copy GB/s (read+write)', 2*(1<<30)*n/ms/1e6)
for o in out: print(o[:600])
