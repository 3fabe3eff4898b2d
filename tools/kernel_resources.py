"""Per-kernel register / scratch / LDS table of libmjx's gfx950 code object (no GPU needed): hipcc --cuda-device-only, unbundle,
llvm-readelf --notes (amdhsa.kernels metadata), c++filt.   python tools/kernel_resources.py [out.json] [out.md]
("a per-kernel VGPR / scratch table tracked beside the chain profile")"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
out_json = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_kernel_resources.json")
out_md = sys.argv[2] if len(sys.argv) > 2 else os.path.splitext(out_json)[0] + ".md"
with tempfile.TemporaryDirectory() as tmp:
    co, elf = os.path.join(tmp, "mjx.co"), os.path.join(tmp, "mjx.elf")
    pre = os.environ.get("MJX_CO")                      # a device-only object compiled earlier (saves the 80 s compile)
    if pre:
        co = pre
    else:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form",
                               "--cuda-device-only", "-c", "-o", co, os.path.join(ROOT, "mjrl_amd", "csrc", "mjx.hip")])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + co, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
    txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", elf], capture_output=True, text=True, check=True).stdout
rows = []
for k in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, k).group(1)
    rows.append(dict(mangled=g("name"), vgpr=int(g("vgpr_count")), agpr=int(re.match(r"\s*(\d+)", k).group(1)), sgpr=int(g("sgpr_count")),
                     scratch_bytes=int(g("private_segment_fixed_size")), static_lds_bytes=int(g("group_segment_fixed_size")),
                     max_flat_workgroup_size=int(g("max_flat_workgroup_size"))))
names = subprocess.run(["c++filt"], input="\n".join(r["mangled"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    r["kernel"] = re.sub(r"\(.*$", "", n.replace("void ", "").replace("mjx::", ""))
    del r["mangled"]
rows.sort(key=lambda r: r["kernel"])
json.dump(rows, open(out_json, "w"), indent=0)
hot = ("k_fused<64, 64, 1, 8", "k_gemm", "k_lw_head", "k_mlp_fit", "k_policy_fit", "k_bl_gram", "k_cg_", "k_reduce_partials4")
with open(out_md, "w") as f:
    f.write("# Register / scratch / static-LDS use per kernel (gfx950 code object of csrc/mjx.hip, `tools/kernel_resources.py`)\n\n"
            "`vgpr` = architectural VGPRs + AGPRs allocated (the unified file: 512 per lane at one wave per SIMD, 256 at two); `scratch` > 0 = spilled registers "
            "(bytes per lane).  Dynamic LDS (the fused kernels' 154 KB, the GEMMs' 72-144 KB) is set at launch and not in this table.\n\n"
            "| kernel | vgpr (of which agpr) | sgpr | scratch B | static LDS B |\n|---|---|---|---|---|\n")
    for r in rows:
        if any(r["kernel"].startswith(h) for h in hot) or r["scratch_bytes"]:
            f.write("| `%s` | %d (%d) | %d | %d | %d |\n" % (r["kernel"], r["vgpr"], r["agpr"], r["sgpr"], r["scratch_bytes"], r["static_lds_bytes"]))
    f.write("\n%d kernels in all; the full list: `%s`.\n" % (len(rows), os.path.relpath(out_json, ROOT)))
print("kernels:", len(rows), "with scratch:", sum(1 for r in rows if r["scratch_bytes"]))
for r in rows:
    if r["scratch_bytes"]:
        print("  %-70s vgpr %3d scratch %4d B" % (r["kernel"][:70], r["vgpr"], r["scratch_bytes"]))
