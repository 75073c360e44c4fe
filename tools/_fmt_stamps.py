import sys,re
lab=None
for line in sys.stdin:
    if line.startswith('GR_ABL'): lab=line.strip(); continue
    if 'GR_STAMP' not in line: continue
    v=[float(x.split('/')[0]) for x in re.findall(r'\] ([0-9.]+/[0-9.]+)', line)]
    print(lab, 'entry %.1f prologue %.1f |' % (v[32], v[0]-v[32]), 'gin0 %.1f | L1 %.1f L2 %.1f L3 %.1f L4 %.1f L5 %.1f | bnd0 %.1f (%.1f %.1f %.1f) | final %.1f | bnd1: fold %.1f lds+atomics %.1f wload %.1f barrier %.1f bn %.1f | total %.1f' % (v[4]-v[1], v[8]-v[7], v[12]-v[11], v[16]-v[15], v[20]-v[19], v[24]-v[23], v[7]-v[4], v[5]-v[4], v[6]-v[5], v[7]-v[6], v[31]-v[30], v[2]-v[8], v[3]-v[2], v[9]-v[3], v[10]-v[9], v[11]-v[10], v[31]))
