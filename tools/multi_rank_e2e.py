import os, subprocess, time, numpy as np, shutil, sys
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
CLI=os.path.join(ROOT,"squishrs_amd","host","squishrs")
base="/dev/shm/sqm"; shutil.rmtree(base, ignore_errors=True); os.makedirs(base+"/in")
rng=np.random.default_rng(5)
t0=time.time()
for i in range(64):
    rng.bytes(64<<20)  # warm
    break
for i in range(64):
    open(f"{base}/in/f{i:03d}.bin","wb").write(rng.bytes(64<<20))
print("tree 4 GiB written in %.1f s"%(time.time()-t0), flush=True)
for args in (["pack"], ["pack","--devices","0"], ["pack","--devices","0,0"], ["pack","--devices","0,0,0,0"]):
    out=base+"/o_%d.squish"%len(args[-1])
    t0=time.time()
    r=subprocess.run([CLI,*args,base+"/in","-o",out],capture_output=True,text=True,env=dict(os.environ,SQUISH_TIMING="1"))
    dt=time.time()-t0
    print(args, "rc",r.returncode,"%.2f s = %.0f MiB/s"%(dt,4096/dt), flush=True); print("   "+r.stderr.strip().replace("\n","\n   ")[-900:], flush=True)
a=open(base+"/o_4.squish","rb").read(); 
for n in (1,3,7):
    b=open(base+"/o_%d.squish"%n,"rb").read()
    print(n, "equal to one-GPU archive:", a[19:]==b[19:], len(b))
shutil.rmtree(base, ignore_errors=True)
