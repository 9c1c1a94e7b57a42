#!/bin/bash
# round 6, evidence set r06_t4 (the round's final tree): tools/collect_evidence.sh
bash tools/collect_evidence.sh r06_t4 2>&1 | tail -120
