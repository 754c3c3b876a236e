#!/bin/bash
# the judged artefacts of the current build + the full GPU suite, one GPU session
TAG=${1:-r04f}
bash benchmarks/r04_artifacts.sh $TAG
bash benchmarks/r04_t.sh
