cd $GRAFT_REPO_ROOT
bash tools/ab.sh comp "c10_tail1||--law compressible --steps 40" "c10_tail0|JH_TAIL_REDUCE=0|--law compressible --steps 40" "c5||--law compressible --steps 40 --cells 5000000" "c10_dt1||--law compressible --steps 40 --dt 1.0" 2>&1 | cut -c1-400
