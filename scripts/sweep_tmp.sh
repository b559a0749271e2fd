echo -n "long bricks: "; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c60-180
echo -n "cubic bricks: "; DLKA_GX_BRICK=cube python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c60-180
python scripts/prof_op.py --C 32 --N 32 --iters 20 --ops deform_bwd_input
