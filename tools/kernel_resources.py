#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch / static LDS of every kernel in libartist_hip.so (from the code object's metadata notes).
usage: python tools/kernel_resources.py [substring]"""
import pathlib, re, subprocess, sys, tempfile
ROOT = pathlib.Path(__file__).resolve().parent.parent
llvm = pathlib.Path("/opt/rocm/lib/llvm/bin")
lib = pathlib.Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "artist_amd" / "libartist_hip.so"
want = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as tmp:
    tmp = pathlib.Path(tmp)
    fat = tmp / "fat.bin"
    subprocess.run([str(llvm / "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", str(lib), str(tmp / "s.so")], check=True)
    blob = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    for k, st in enumerate(starts):
        part = tmp / f"b{k}.bin"
        part.write_bytes(blob[st: starts[k + 1] if k + 1 < len(starts) else len(blob)])
        code = tmp / f"c{k}.co"
        subprocess.run([str(llvm / "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={part}", f"--output={code}"], check=True)
        if not code.exists() or code.stat().st_size == 0:
            continue
        notes = subprocess.run([str(llvm / "llvm-readelf"), "--notes", str(code)], capture_output=True, text=True).stdout
        for block in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            if not name or want not in name.group(1):
                continue
            dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
            get = lambda key: (re.search(rf"\.{key}:\s+(\d+)", block) or [0, "?"])[1]
            print(f"{dem[:110]:110s} vgpr {get('vgpr_count'):>3s} sgpr {get('sgpr_count'):>3s} scratch {get('private_segment_fixed_size'):>5s} "
                  f"lds {get('group_segment_fixed_size'):>6s} spills v{get('vgpr_spill_count')} s{get('sgpr_spill_count')}")
