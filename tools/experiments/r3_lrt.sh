cd "$(dirname "$0")/../.."
for lib in "$@"; do
  export BASEVAR_AMD_LIB=$PWD/basevar_amd/lib/$lib
  echo "== $lib"
  tools/trace_timeline.sh v$lib --samples 10000 --batch-sites 100000 2>/dev/null | grep -E "lrt16|tail16" | tail -4
done
