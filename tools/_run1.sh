for rb in 9 5 3; do echo RB=$rb; RK_PW16_RB=$rb timeout 300 python tools/pw_bf16_time.py 256,288,288,14,14 256,144,144,28,28 2>&1 | grep "/pk"; done
