#!/bin/bash
# Round 6, second GPU session: kernel stats, timelines and PMC traffic of every workload at HEAD --
# the measurement pass of tools/r5_final.sh without its test run (tools/r6_first.sh ran the tests):
#   tools/r6_prof.sh [TAG=r06b]
# Everything lands in gpurun_out/TAG; copy what is to be judged to profiles/r06_*.
exec bash "$(dirname "$0")/r5_final.sh" "${1:-r06b}" notests
