"""Where the spilled registers of libmjx's kernels are touched (no GPU needed): every scratch_load / scratch_store of the gfx950 ISA is
attributed to the INNERMOST loop (backward branch) that contains it, with that loop's MFMA count -- a spill in set-up code costs
nothing, one inside a matrix loop does.   python tools/scratch_by_loop.py [mjx.s] > profiles/r06_kernel_scratch_by_loop.txt
(the .s: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form --cuda-device-only -S mjrl_amd/csrc/mjx.hip; 80 s)"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else None
if path is None:
    path = os.path.join(tempfile.mkdtemp(), "mjx.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form",
                           "--cuda-device-only", "-S", "-o", path, os.path.join(ROOT, "mjrl_amd", "csrc", "mjx.hip")])
funcs, cur = {}, None
for line in open(path):
    m = re.match(r"^(_ZN3mjx\S+):\s", line)
    if m:
        cur = m.group(1); funcs[cur] = []; continue
    if cur is not None:
        funcs[cur].append(line)
names = list(funcs)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
print("scratch (spill) instructions of libmjx's gfx950 kernels by the INNERMOST loop that contains them (loop = backward branch in the ISA;")
print("hipcc -O3 -mllvm -amdgpu-mfma-vgpr-form).  Kernels that do not appear have no scratch instruction at all (tools/kernel_resources.py: scratch_bytes 0).\n")
for n, d in zip(names, dem):
    body = funcs[n]
    sc = [i for i, l in enumerate(body) if re.search(r"\bscratch_(load|store)", l)]
    if not sc:
        continue
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\S+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    groups = collections.Counter()
    for i in sc:
        enc = [(b - a, a, b) for a, b in loops if a <= i <= b]
        if not enc:
            groups[("outside any loop (set-up / tail code)", 0, 0)] += 1
            continue
        _, a, b = min(enc)
        groups[("a loop of %d instructions with %d MFMAs" % (b - a, sum(1 for l in body[a:b] if "v_mfma" in l)), a, b)] += 1
    print(re.sub(r"\(.*$", "", d.replace("void ", "").replace("mjx::", "")))
    for (desc, a, b), cnt in sorted(groups.items(), key=lambda x: x[0][1]):
        print("    %3d scratch instructions in %s" % (cnt, desc))
