timeout 300 python tools/pw16_check.py 2>&1 | grep -c OK; timeout 300 python tools/pw16_check.py 2>&1 | grep FAIL | head -3
for d in 0 8; do echo dbg=$d; RK_PW16_DBG=$d timeout 300 python tools/pw_bf16_time.py 2>&1 | grep "/pk"; done
