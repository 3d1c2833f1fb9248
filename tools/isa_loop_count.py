"""Counts the instructions of the innermost period loop of a paths_kernel variant in a
gfx950 .s file and weights them by the measured issue costs relative to a plain VALU op
(profiles/r01/ubench_instruction_rates.txt, 8 waves/SIMD, warmed up)."""
import sys
from collections import Counter
W = {'v_pk_fma_f32': 1.9, 'v_pk_mul_f32': 1.9, 'v_pk_add_f32': 1.9, 'v_mad_u64_u32': 2.29, 'v_mul_hi_u32': 1.78,
     'v_mul_lo_u32': 1.78, 'v_sqrt_f32_e32': 3.55, 'v_cvt_f32_u32_e32': 1.8, 'v_cvt_f32_i32_e32': 1.8,
     'v_rcp_f32_e32': 3.55, 'v_rsq_f32_e32': 3.55}
path, variant = sys.argv[1], sys.argv[2]
periods = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lines = open(path).read().splitlines()
beg = [i for i, l in enumerate(lines) if l.startswith('_ZN4smmc12_GLOBAL__N_112paths_kernel' + variant)][0]
fin = [i for i, l in enumerate(lines) if i > beg and 's_endpgm' in l][0]
body = lines[beg:fin]
start = [i for i, l in enumerate(body) if 'Inner Loop Header: Depth=2' in l][0]
end = [i for i, l in enumerate(body) if i > start and 's_cbranch_scc' in l][0]
ops = [l.split()[0] for l in body[start:end + 1] if l.strip() and l.strip()[0] not in ';.']
c = Counter(ops)
valu = sum(n for k, n in c.items() if k.startswith('v_'))
tot = sum(n * W.get(k, 1.0) for k, n in c.items() if k.startswith('v_') or k.startswith('ds_'))
print(f"{variant}: {valu} VALU + {sum(n for k, n in c.items() if k.startswith('ds_'))} LDS per {periods} periods "
      f"({valu / periods:.2f} VALU/period); weighted {tot:.1f} -> {tot / periods:.1f} issue units per path-period")
print("  ", dict(c.most_common(14)))
