"""Table of hipcc -Rpass-analysis=kernel-resource-usage output (stderr file): kernel, VGPRs, AGPRs, SGPRs, spills, LDS, occupancy."""
import re
import sys

rows, cur = [], {}
for ln in open(sys.argv[1]):
    m = re.search(r'remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass', ln) or re.search(r':\d+:\d+:\s+(?:remark:\s+)?(.*?)\s+\[-Rpass', ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:') or t.startswith('Name:'):
        if cur:
            rows.append(cur)
        cur = {'name': t.split(':', 1)[1].strip()}
    elif ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for r in rows:
    n = r['name']
    m = re.search(r'conv1d_mfma_(f32|bf16)ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)', n)
    short = f'conv_{m.group(1)}<K{m.group(2)},MT{m.group(3)},NTL{m.group(4)},WM{m.group(5)},WN{m.group(6)},E{m.group(7)}>' if m else n[:60]
    if flt and flt not in short:
        continue
    print(f"{short:44s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sgpr {r.get('TotalSGPRs', r.get('SGPRs','?')):>3s} "
          f"spill {r.get('VGPRs Spill', r.get('VGPR Spill','?')):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?')}")
