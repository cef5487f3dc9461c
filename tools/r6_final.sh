# Round 6, last calls: the profile recipe on the final library (one gpurun call), then the whole GPU suite and smoke() as the driver
# runs them and four soaks (tools/r6_final_rest.sh: a second call — together they do not fit one call's 20 minutes).
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
bash tools/final_profile.sh r6 > gpurun_out/r6_final_profile.log 2>&1; echo "profile rc=$?"; tail -2 gpurun_out/r6_final_profile.log
