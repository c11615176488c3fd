import re,collections,sys
lines=open(sys.argv[1]).read().split('\n')
cur='entry';cnt=collections.Counter();segs=[]
for l in lines:
    m=re.match(r'^(\.LBB\S+):',l)
    if m:
        segs.append((cur,cnt));cur=m.group(1);cnt=collections.Counter();continue
    t=l.strip().split()
    if not t or t[0].startswith(('.',';','//')): continue
    op=t[0]
    if 'dpp' in l or 'quad_perm' in l: op=op+'(dpp)'
    cnt[op]+=1
segs.append((cur,cnt))
for name,c in segs:
    tot=sum(c.values()); v=sum(n for o,n in c.items() if o.startswith('v_'))
    print(name,tot,'valu',v)
    print('   ',c.most_common(16))
