# CPU-only sanitizer build of the oracle (tests/sanitize/test_sanitizers.py replays goldens through it under LD_PRELOAD=libasan):
# both precisions, AddressSanitizer + UndefinedBehaviorSanitizer, any finding aborts.
# Lives under tests/sanitize/ (listed in .gpurunignore): nothing here ever travels to, or runs on, a GPU box.
#   make -C oracle -f ../tests/sanitize/oracle_asan.mk
CC ?= gcc
OUT := _build
SANFLAGS := -O1 -g -fPIC -fopenmp -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer
$(OUT)/libmvrl_oracle_asan.so: mvrl_oracle.c ../include/mvrl.h
	@mkdir -p $(OUT)
	$(CC) $(SANFLAGS) -DREAL=double -DSUF=_f64 -c $< -o $(OUT)/oracle_f64_asan.o
	$(CC) $(SANFLAGS) -DREAL=float -DSUF=_f32 -c $< -o $(OUT)/oracle_f32_asan.o
	$(CC) -shared -fopenmp -fsanitize=address,undefined -o $@ $(OUT)/oracle_f64_asan.o $(OUT)/oracle_f32_asan.o -lm
