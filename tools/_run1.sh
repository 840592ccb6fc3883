timeout 300 python tools/pw16_check.py 2>&1 | grep -c OK; timeout 300 python tools/pw16_check.py 2>&1 | grep "FAIL\|rror" | head -5
timeout 300 python tools/pw_bf16_time.py 2>&1 | grep "wgrad/16"
