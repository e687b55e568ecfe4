# same-box A/B of the dense-batch changes: dense (base / F), row log (base / new), C5 (base / new)
cd $GRAFT_REPO_ROOT
bash tools/gpu/c3dense.sh libfwgym_c3dense_base.so libfwgym_c3dense_F.so
bash tools/gpu/c3dev.sh libfwgym_c3log_base.so libfwgym_c3log_new.so
bash tools/gpu/c5dev.sh libfwgym_c5base.so libfwgym_c5new.so
