#!/bin/bash
# round 6, run 1: the hand-off repro (tools/microbench/handoff.hip) alone, next to a second process, next to the pipeline
bash tools/microbench/run_handoff.sh
