#!/bin/bash
# round 5, call u: the new match kernel on other shapes of input (2 GiB each): no cliff on zeros / short periods / random bytes?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/enc_data_probe.py 2>&1 | tail -12
