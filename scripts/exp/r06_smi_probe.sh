cd $GRAFT_REPO_ROOT
for d in /sys/class/drm/card*/device; do echo $d; ls $d/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo; for f in power1_average power1_input power1_cap power1_cap_max freq1_input; do for h in $d/hwmon/*; do [ -r $h/$f ] && echo "$f = $(cat $h/$f)"; done; done; cat $d/pp_dpm_sclk 2>/dev/null | head -5; done 2>&1 | head -60
time rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -30
which amd-smi; python -c "import amdsmi" 2>&1 | head -2
