"""Copy the summaries tools/regen_profiles.sh left under gpurun_out/prof_<round>/ into profiles/ (tracked):
    python tools/collect_profiles.py r02"""
import glob, os, shutil, subprocess, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{R}")
dst = os.path.join(root, "profiles")


def stats(sub, name):
    fs = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if not fs:
        print("missing", sub)
        return
    shutil.copy(fs[0], os.path.join(dst, f"{R}_{name}_kernel_stats.csv"))
    print("wrote", f"{R}_{name}_kernel_stats.csv")


stats("roofline", "roofline_probe")
stats("decode", "decode")
stats("bench", "bench")
stats("paligemma", "paligemma_decode")
stats("prefill", "paligemma_prefill")
stats("vlm", "vlm")
stats("train", "train_step")
dirs = [os.path.join(src, d) for d in ("pmc_fetch", "pmc_write", "pmc_hit")]
if all(os.path.isdir(d) for d in dirs):
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gemm_pmc_json.py")] + dirs, capture_output=True, text=True)
    if out.returncode == 0:
        open(os.path.join(dst, f"{R}_gemm_pmc.json"), "w").write(out.stdout)
        print("wrote", f"{R}_gemm_pmc.json")
    else:
        print(out.stderr[-2000:])
tl = os.path.join(src, "decode_timeline.log")
if os.path.exists(tl):
    keep = [l for l in open(tl).read().splitlines() if l[:1] in " lf" and ("|" in l or "stamped" in l)]
    open(os.path.join(dst, f"{R}_decode_timeline.txt"), "w").write("\n".join(keep) + "\n")
    print("wrote", f"{R}_decode_timeline.txt")
for log in ("roofline.log", "decode.log", "paligemma.log", "prefill.log", "vlm.log", "train.log"):
    p = os.path.join(src, log)
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines()
                 if l.startswith(("{", "decode step", "PaliGemma shape", "prefill + first token", "caption training", "train_only", "losses"))]
        if lines:
            open(os.path.join(dst, f"{R}_{log.replace('.log', '')}_run.txt"), "w").write("\n".join(lines) + "\n")
