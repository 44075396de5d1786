#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests -m gpu -q -x -k "dropout_of_the_layer_in_front" 2>&1 | tail -4
