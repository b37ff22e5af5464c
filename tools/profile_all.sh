#!/bin/bash
# Runs ON THE GPU BOX: the four BASELINE configurations through tools/profile_gpu.sh (stats pass + PMC passes each).
# (Cornell and AO at reduced spp: the kernels' per-launch behaviour does not depend on it, a 1024 / 4096 spp frame would take minutes per pass.)
set -e
tools/profile_gpu.sh bunny python3 bench.py --steps 3 --warmup 1 --no-cpu --no-others
tools/profile_gpu.sh cornell python3 bench.py --workload cornell --spp 256 --steps 2 --warmup 1 --no-cpu
tools/profile_gpu.sh grid python3 bench.py --workload grid --steps 2 --warmup 1 --no-cpu
tools/profile_gpu.sh ao python3 bench.py --workload ao --spp 16 --steps 2 --warmup 1 --no-cpu
