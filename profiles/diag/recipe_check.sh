# One seed, 5 M steps of each reference recipe: a quick look that the tree still learns as
# results/README.md records (cramped_room ~25 at 5 M, forced_coordination ~1-2).
SEEDS=1 STEPS=5000000 bash results/run_reference_recipes.sh 2>&1 | tail -4
python - <<'PY'
import glob
import numpy as np
for d in ("r02_makefile_cramped", "r02_readme_cramped", "r02_makefile_forced"):
    for f in sorted(glob.glob("gpurun_out/results/%s/*return*" % d))[:1]:
        x = np.loadtxt(f, skiprows=1)
        print(d, len(x), "episodes, mean return of the last 10: %.2f" % x[-10:].mean())
PY
