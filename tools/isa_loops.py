"""Loop statistics of one kernel in a `hipcc -S` listing (instruction classes per back edge): python tools/isa_loops.py
k.s <line of the kernel label>.  Used to count the instructions of the fill kernels' steady-state loops."""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
start=int(sys.argv[2]); 
# kernel end: s_endpgm after start
end=start
while not lines[end].strip().startswith('.Lfunc_end'): end+=1
lab={}
for i in range(start,end):
    m=re.match(r'^(\.LBB\d+_\d+):',lines[i])
    if m: lab[m.group(1)]=i
loops=[]
for i in range(start,end):
    m=re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)',lines[i]) or re.match(r'\s+s_branch\s+(\.LBB\d+_\d+)',lines[i])
    if m and m.group(1) in lab and lab[m.group(1)]<i:
        loops.append((lab[m.group(1)],i))
for a,b in loops:
    body=[l.strip() for l in lines[a:b+1] if l.strip() and not l.strip().startswith((';','.'))]
    inner=not any(a<c and d<b for c,d in loops)
    cnt={}
    for l in body:
        op=l.split()[0]
        k='valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'ds' if op.startswith('ds_') else 'mem'
        cnt[k]=cnt.get(k,0)+1
    print(a+1,b+1,len(body),'inner' if inner else 'outer',cnt, 'max3=%d'%sum('v_max3' in l for l in body))
