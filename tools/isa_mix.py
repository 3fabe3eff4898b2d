"""Instruction mix of a kernel's hot loop from the library's assembly (hipcc -S --cuda-device-only of csrc/mjx.hip):
python tools/isa_mix.py <mjx.s> <mangled-name-substring> -- the largest backward-branch span of the kernel = its tile / step loop.
Counts what fp32 MFMAs do NOT hide: vector-ALU instructions by opcode, AGPR moves, s_nop."""
import collections, re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_ZN3mjx\w*:", l) and pat in l][0]
end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
best = (0, 0, 0)
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
        best = (i - labels[m.group(1)], labels[m.group(1)], i)
loop = [l.strip().split()[0] for l in body[best[1]:best[2] + 1] if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
c = collections.Counter(loop)
mf = sum(v for o, v in c.items() if "mfma" in o)
va = sum(v for o, v in c.items() if o.startswith("v_") and "mfma" not in o)
print("%s: loop of %d instructions: %d MFMA, %d vector-ALU (%d accvgpr moves, %d v_add_u32, %d v_lshl_add_u64 / v_mad_u64), %d DS, %d s_nop, %d s_waitcnt"
      % (lines[start].split(":")[0][:90], len(loop), mf, va, sum(v for o, v in c.items() if "accvgpr" in o), c["v_add_u32_e32"],
         c["v_lshl_add_u64"] + c["v_mad_u64_u32"] + c["v_mad_i64_i32"], sum(v for o, v in c.items() if o.startswith("ds_")), c["s_nop"], c["s_waitcnt"]))
print("  " + ", ".join("%d %s" % (v, k) for k, v in c.most_common(28) if k.startswith("v_") and "mfma" not in k))
