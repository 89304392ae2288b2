# sourced by the scripts/ab_*.sh A/B helpers (hand-built objects with experiment flags).
#  * per-file production flags come from pixparse_amd/build.py (EXTRA_FLAGS), so that an A/B measures production codegen + the knob;
#  * whatever happens, the DEFAULT build is restored on exit: an experiment object left in csrc/ would be newer than its source, and
#    the mtime-based incremental build would link it into the next library (wrong-results timing flags left behind).
export PIXPARSE_AMD_SKIP_BUILD_CHECK=1
AB_ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
extra_flags() { python -c "import sys; sys.path.insert(0, '$AB_ROOT'); from pixparse_amd import build; print(' '.join(build.EXTRA_FLAGS.get('$1', [])))"; }
restore_default() {
  ( cd "$AB_ROOT" && unset SPX_DROP SPX_OPTS && python -m pixparse_amd.build --force > /dev/null && echo "default build restored" )
}
trap restore_default EXIT
