#!/bin/bash
# Experiment: cycles per substep inside make_constraints (row table, leading-row fill, contact-row fill, groups, couplings, block inverses)
cd $GRAFT_REPO_ROOT
AVSIM_EXTRA_FLAGS="-DAVSIM_PROBE_ROWS" python -m av_aloha_amd.build --force > /dev/null 2>&1
for t in "slot_insertion 3" "sew_needle 3" "hook_package 2"; do set -- $t; echo "$1: $(TASK=$1 ARMS=$2 python tools/prof_phases.py 4096 2>/dev/null | grep -E "^  rows|probe slots" | tr '\n' ' ' | cut -c1-330)"; done
