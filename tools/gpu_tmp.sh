#!/bin/bash
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
bash tools/gpu_round_profile.sh r02_e
