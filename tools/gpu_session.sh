# One GPU-box session, parametrised (replaces the per-session gpu_round5_[a-h].sh scripts).  Everything lands under
# gpurun_out/<tag>/.  Usage (through gpurun):  bash tools/gpu_session.sh <tag> <step> [<step> ...]
#   tests            python -m pytest tests -m gpu -x -q
#   tests:<expr>     the same with -k <expr>
#   bench20          python bench.py --steps 20 --warmup 5      (the driver's command)   -> bench_k20.{log,json}
#   bench200         python bench.py                                                     -> bench_k200.{log,json}
#   extras20         python bench.py --steps 20 --warmup 5 --extras                      -> bench_k20_extras.{log,json}
#   prof:<name>:<bench args, '+' for spaces>    tools/gpu_profile.sh <name> <args>       -> gpurun_out/prof_<name>/
#   py:<script>[:<args, '+' for spaces>]        python <script> <args> > <tag>/<script basename>.txt
#   lib:<variant.so>                            export JSSENV_AMD_LIB for the following steps (variants/ built here, shipped with the tree)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for step in "$@"; do
  case $step in
    tests)    timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log ;;
    tests:*)  timeout 1500 python -m pytest tests -m gpu -x -q -k "${step#tests:}" > $O/pytest_k.log 2>&1; tail -5 $O/pytest_k.log ;;
    bench20)  timeout 600 python bench.py --steps 20 --warmup 5 --detail $O/bench_k20.json > $O/bench_k20.log 2>&1; tail -1 $O/bench_k20.log ;;
    bench200) timeout 600 python bench.py --detail $O/bench_k200.json > $O/bench_k200.log 2>&1; tail -1 $O/bench_k200.log ;;
    extras20) timeout 900 python bench.py --steps 20 --warmup 5 --extras --detail $O/bench_k20_extras.json > $O/bench_k20_extras.log 2>&1; tail -1 $O/bench_k20_extras.log ;;
    prof:*)   rest=${step#prof:}; name=${rest%%:*}; a=${rest#*:}; bash tools/gpu_profile.sh $name ${a//+/ } > $O/prof_$name.log 2>&1; tail -30 $O/prof_$name.log ;;
    py:*)     rest=${step#py:}; scr=${rest%%:*}; a=""; [ "$rest" != "$scr" ] && a=${rest#*:}; out=$O/$(basename $scr .py)${a:+_${a//+/_}}.txt; timeout 900 python $scr ${a//+/ } > $out 2>&1; tail -40 $out ;;
    lib:*)    export JSSENV_AMD_LIB=$R/${step#lib:} ;;
    *)        echo "unknown step $step" ;;
  esac
done
