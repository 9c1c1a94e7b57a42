#!/bin/bash
# round 6, evidence set r06_t3 (tree 3ad...: fused apply passes, batched weight-gradient forks, wider pack launches): tools/collect_evidence.sh
bash tools/collect_evidence.sh r06_t3
