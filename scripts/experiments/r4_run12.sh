#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_band.py -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed rc=$rc"; tail -60 $O/pytest.log; exit $rc; }
timeout -k 10 300 python scripts/experiments/exp_band_chunks.py > $O/band_chunks.json 2> $O/band_chunks.err; echo rc=$?; cat $O/band_chunks.json; tail -3 $O/band_chunks.err
