cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('SMOKE OK')" 2>&1 | tail -5
