"""Resource usage of every kernel in a built liblmpc_hip.so (code-object notes): VGPRs, AGPRs, SGPRs, scratch, spills.
usage: kinfo.py [lib.so]"""
import subprocess, sys, re, tempfile, os
lib = sys.argv[1] if len(sys.argv) > 1 else "racing-lmpc-ros2_amd/lib/liblmpc_hip.so"
LL = "/opt/rocm/lib/llvm/bin/"
d = tempfile.mkdtemp()
out = os.path.join(d, "co")
fb = os.path.join(d, "fb")
subprocess.check_call([LL + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb])
# one offload bundle per translation unit of the library (lmpc_lib.hip, lmpc_lib_minreg.hip), back to back in the section
blob = open(fb, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
notes = ""
for n, a in enumerate(starts):
    part = os.path.join(d, "fb%d" % n)
    open(part, "wb").write(blob[a:starts[n + 1] if n + 1 < len(starts) else len(blob)])
    r = subprocess.run([LL + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out + str(n)], capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr); sys.exit(1)
    notes += subprocess.run([LL + "llvm-readelf", "--notes", out + str(n)], capture_output=True, text=True).stdout
blocks = re.split(r"\n\s*- \.agpr_count:", notes)
for b in blocks[1:]:
    g = lambda key: (re.search(r"\." + key + r":\s*(\S+)", b) or [None, "?"])[1]
    name = g("name")
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void ", "")
    agpr = re.match(r"\s*(\d+)", b).group(1)
    print(f"{dem:60s} vgpr {g('vgpr_count'):>4} agpr {agpr:>4} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} vspill {g('vgpr_spill_count'):>4} sspill {g('sgpr_spill_count'):>4}")
