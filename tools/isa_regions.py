"""static instruction counts between the PFK_MARK comments of one kernel of an ISA listing: python isa_regions.py <file.s> <mangled-substring>"""
import re, sys
txt = open(sys.argv[1]).read()
sub = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(sub) + r'\S*):', txt, re.M)
i = m.start(); j = txt.index('.Lfunc_end', i)
body = txt[i:j].splitlines()
cur = '00_prologue'; counts = {}; order = []
for ln in body:
    mm = re.search(r'; PF[K]?_MARK (\S+)', ln)
    if mm:
        cur = mm.group(1)
        if cur not in order: order.append(cur)
        continue
    t = ln.strip()
    if not t or t.startswith(('.', ';', '/')) or t.endswith(':'): continue
    op = t.split()[0]
    c = counts.setdefault(cur, {'v': 0, 's': 0, 'ds': 0, 'mem': 0, 'v64': 0, 'trans': 0})
    if op.startswith('v_'):
        c['v'] += 1
        if '_f64' in op: c['v64'] += 1
        if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', op): c['trans'] += 1
    elif op.startswith('s_'): c['s'] += 1
    elif op.startswith('ds_'): c['ds'] += 1
    elif op.startswith(('buffer_', 'global_', 'flat_')): c['mem'] += 1
print(m.group(1))
print("region (instructions AFTER this mark, static)     VALU (f64, transcendental)  SALU  LDS  VMEM")
tot = 0
for k in ['00_prologue'] + order:
    c = counts.get(k)
    if c:
        print(f"{k:28s} {c['v']:5d} ({c['v64']:3d}, {c['trans']:3d}) {c['s']:5d} {c['ds']:4d} {c['mem']:4d}")
