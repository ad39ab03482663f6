"""Stamp a profiles/*_pmc_counters.json with the commit count of the tree it was collected on (the GPU boxes have no .git):
python tools/stamp_pmc.py profiles/r05_x_pmc_counters.json   -- run BEFORE committing the file; bench.py reports
roofline.counters.pmc_age_commits = commits since then."""
import json, subprocess, sys
n = int(subprocess.check_output(['git', 'rev-list', '--count', 'HEAD']).decode())
for f in sys.argv[1:]:
    d = json.load(open(f))
    d['git_commit_count'] = n + 1          # the commit that adds the file
    json.dump(d, open(f, 'w'), indent=1)
    print(f, '-> git_commit_count', n + 1)
