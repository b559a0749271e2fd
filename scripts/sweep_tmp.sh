for t in 256 512 1024; do echo -n "gx threads=$t: "; DLKA_GX_THREADS=$t python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c60-180; done
