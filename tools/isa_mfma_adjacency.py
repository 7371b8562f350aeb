import re,collections,sys
lines=open(sys.argv[1]).read().splitlines()
pat=sys.argv[2] if len(sys.argv)>2 else 'Lb0EEEvNS_5KArgsE'
start=[i for i,l in enumerate(lines) if l.startswith('_ZN6nerfds18render_rays_kernel') and pat in l][0]
end=[i for i in range(start,len(lines)) if lines[i].strip().startswith('s_endpgm')][0]
prev=None; gap=0; hist=collections.Counter()
n=0
for i in range(start,end):
    l=lines[i].strip()
    if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'): continue
    m=re.match(r'v_mfma_\S+ (\S+), (\S+), (\S+), (\S+)',l)
    if m:
        n+=1
        dst=m.group(1).rstrip(',')
        if prev is not None: hist[(dst==prev, 'adjacent' if gap==0 else 'fillers')]+=1
        prev=dst; gap=0
    else: gap+=1
print(sys.argv[1], 'mfma', n, {f'{k[0] and "same" or "diff"}-{k[1]}': v for k,v in sorted(hist.items())})
