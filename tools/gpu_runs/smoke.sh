#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
time python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
