import re, itertools, sys
s=open('/tmp/igemm.s').read()
name=sys.argv[1] if len(sys.argv)>1 else '_Z19afi_pix_gemm_kernelILi128ELi128ELi2ELi2ELb0EEv10AfiPixGemmiii'
i=s.index(name+':'); j=s.index('.Lfunc_end', i)
body=s[i:j].split('\n')
idx=[n for n,l in enumerate(body) if 'v_mfma' in l]
lo,hi=idx[0]-80, idx[-1]+10
def cls(o):
    if o.startswith('v_mfma'): return 'M'
    if o.startswith('ds_read'): return 'DSR'
    if o.startswith('ds_write'): return 'DSW'
    if o.startswith('global_load'): return 'GLD'
    if o=='s_barrier': return 'BAR'
    if o=='s_waitcnt': return 'WAIT'
    if o.startswith('s_cbranch') or o.startswith('s_branch'): return 'BR'
    if o.startswith('v_'): return 'v'
    return 's'
out=[]
for l in body[lo:hi]:
    t=l.strip()
    if not t or t.startswith(';'): continue
    if re.match(r'\.LBB\d+_\d+:',t): out.append('\n'+t.split(':')[0]+':'); continue
    if t.startswith('.'): continue
    op=t.split()[0]
    c=cls(op)
    if c=='WAIT': c='W('+t.split(None,1)[1].replace(' ','')+')'
    if c=='BR': c='BR('+t.split()[-1]+')'
    out.append(c)
res=[]
for k,g in itertools.groupby(out):
    n=len(list(g)); res.append(k if n==1 else f"{k}{n}")
print(' '.join(res))
