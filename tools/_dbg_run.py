import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools"))
import fuzz_api as F
text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dbg_reduced.da")).read()
F.run(2605911, True, lambda t: text)
