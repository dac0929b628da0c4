import sys,re,subprocess
src=sys.argv[1]
out=subprocess.run(["/opt/rocm/bin/hipcc","-O3","-std=c++17","--offload-arch=gfx950","-ffp-contract=off","-Rpass-analysis=kernel-resource-usage","-c",src,"-o","/tmp/exp/st/x.o"],capture_output=True,text=True).stderr
cur=None
for line in out.splitlines():
    m=re.search(r"remark: (.*?): (.*?) \[",line) 
    m2=re.search(r"Function Name: (\S+)",line)
    if m2:
        name=subprocess.run(["c++filt",m2.group(1)],capture_output=True,text=True).stdout.strip()
        name=re.sub(r"\(.*","",name); cur={'name':name}; continue
    m3=re.search(r"remark: [^:]*:\d+:\d+:\s+(\w[\w ]*?): (\S+)",line)
    if m3 and cur is not None:
        cur[m3.group(1).strip()]=m3.group(2)
        if m3.group(1).strip().startswith('LDS Size'):
            print(f"{cur['name'][:70]:70s} VGPR {cur.get('VGPRs')} AGPR {cur.get('AGPRs')} SGPR {cur.get('TotalSGPRs')} scratch {cur.get('ScratchSize [bytes/lane]')} occ {cur.get('Occupancy [waves/SIMD]')} LDS {cur.get('LDS Size [bytes/block]')}")
