#!/bin/bash
out=gpurun_out/r03_n
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $out/pytest.log 2>&1
tail -6 $out/pytest.log
