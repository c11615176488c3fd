"""Split a kernel's ISA at '; MARK name' comments and count VALU instructions per named section."""
import re,sys,collections
lines=open(sys.argv[1]).read().split('\n')
cur='(pre)';tot=collections.Counter();ops=collections.defaultdict(collections.Counter)
for l in lines:
    m=re.search(r'; MARK (\S+)',l)
    if m: cur=m.group(1);continue
    t=l.strip().split()
    if not t or not t[0].startswith(('v_','s_nop')): continue
    op=t[0]+('(dpp)' if ('quad_perm' in l or 'row_' in l) else '')
    tot[cur]+=1;ops[cur][op]+=1
tt=sum(tot.values())
for k,v in sorted(tot.items(),key=lambda x:-x[1]):
    print('%-14s %6d  %5.1f%%  %s'%(k,v,100*v/tt,ops[k].most_common(6)))
print('total',tt)
