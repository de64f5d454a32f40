import re, collections, sys, subprocess
src='kajo_amd/csrc/kernel_fast.hip' if len(sys.argv)<2 else sys.argv[1]
kern='kajo_render_fast' if len(sys.argv)<3 else sys.argv[2]
extra=sys.argv[3:] 
subprocess.run(['hipcc','--offload-arch=gfx950','-O3',('-ffp-contract=off' if 'strict' in src else '-ffp-contract=fast'),'-fno-slp-vectorize','-std=c++17','-gline-tables-only','-Iinclude','-Ikajo_amd/csrc','-S','--cuda-device-only',src,'-o','/tmp/t/blk.s']+extra,check=True,stderr=subprocess.DEVNULL)
lines=open('/tmp/t/blk.s').read().split('\n')
start=[i for i,l in enumerate(lines) if l.startswith(kern+':')][0]
end=[i for i,l in enumerate(lines) if i>start and l.startswith('.Lfunc_end')][0]
files={}
for l in lines:
    m=re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s*"?([^"]*)"?',l)
    if m: files[int(m.group(1))]=(m.group(3) or m.group(2)).split('/')[-1]
blocks=[]; cur={'label':'entry','valu':0,'salu':0,'mem':0,'trans':0,'locs':collections.Counter()}
loc=None
for l in lines[start:end]:
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)',l)
    if m: loc=(files.get(int(m.group(1)),'?')[:10],int(m.group(2))); continue
    m=re.match(r'(\.LBB\d+_\d+):',l)
    if m:
        blocks.append(cur); cur={'label':m.group(1),'valu':0,'salu':0,'mem':0,'trans':0,'locs':collections.Counter()}; continue
    t=l.strip()
    if not t or t.startswith(('.',';')) or t.endswith(':'): continue
    op=t.split()[0]
    if op.startswith('v_'):
        cur['valu']+=1; cur['locs'][loc]+=1
        if re.match(r'v_(rcp|rsq|sqrt|sin|cos|exp|log)_',op): cur['trans']+=1
    elif op.startswith('s_') and not op.startswith(('s_nop','s_waitcnt')): cur['salu']+=1
    elif op.startswith(('ds_','global_','buffer_','flat_')): cur['mem']+=1
blocks.append(cur)
tot=0
for b in blocks:
    tot+=b['valu']
    if b['valu']>=10:
        top=', '.join('%s:%d(%d)'%(k[0],k[1],c) for k,c in b['locs'].most_common(5) if k)
        print('%-10s valu %3d (trans %2d) salu %3d mem %2d | %s'%(b['label'],b['valu'],b['trans'],b['salu'],b['mem'],top))
print('total valu',tot, 'salu', sum(b['salu'] for b in blocks))
