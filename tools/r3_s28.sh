cd $GRAFT_REPO_ROOT
for f in 1 0; do echo "== dr_form=$f"; PROXTV_DR_FORM=$f timeout 300 python tools/lambda_probe.py --lams 0.2,0.3,0.4,0.5,0.6,0.65 2>&1 | grep "^mode"; done
