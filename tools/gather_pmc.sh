#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the C2 window-gather launch alone -> gpurun_out/gather_pmc.json
# (copy to profiles/rNN_gather_pmc.json).  One counter per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/gp_$ctr
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/gp_$ctr -o g -- python $ROOT/tools/gather_only.py 6 > $ROOT/gpurun_out/gather_pmc_$ctr.log 2>&1
done
python - "$ROOT" <<'PY'
import hashlib, json, sqlite3, subprocess, sys, glob
root = sys.argv[1]
def counter(tag):
    db = sqlite3.connect(glob.glob(f"/tmp/gp_{tag}/**/*.db", recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(f"select e.value, d.id, d.end - d.start from {pmc} e join {info} i on e.pmc_id=i.id join {disp} d "
                           f"on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id where i.name='{tag}' and "
                           f"s.kernel_name like '%k_window_gather%'"))
    ids = {r[1] for r in rows}
    return sum(r[0] for r in rows) / len(ids), len(ids), sum({r[1]: r[2] for r in rows}.values()) / len(ids)
f, nf, dur = counter("FETCH_SIZE")
w, nw, _ = counter("WRITE_SIZE")
alg = 599976 * 5824
hbm = f * 1024 * 2 + w * 1024
sha = hashlib.sha256(open(f"{root}/deepof_amd/csrc/k_gather.hip", "rb").read()).hexdigest()[:16]
json.dump({"kernel": "k_window_gather (C2 materialisation: 599,976 windows per launch, fp32)",
           "commands": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/gather_only.py 6",
                        "rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python tools/gather_only.py 6"],
           "launches_averaged": min(nf, nw), "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
           "avg_duration_ns_under_pmc": dur,
           "corrections": "MI355X_MICROARCH.md HBM section: FETCH_SIZE tallies wide coalesced reads at half their bytes on "
                          "gfx950 -> doubled (upper bound); WRITE_SIZE uncorrected",
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
           "source_sha": sha, "source": "deepof_amd/csrc/k_gather.hip"}, open(f"{root}/gpurun_out/gather_pmc.json", "w"), indent=1)
print(open(f"{root}/gpurun_out/gather_pmc.json").read())
PY
