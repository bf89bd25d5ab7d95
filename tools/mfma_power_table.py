"""Clock and matrix-pipe occupancy of tools/mfma_power.hip's kernels from one rocprofv3 counter pass:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- ./tools/bin/mfma_power
    python tools/mfma_power_table.py <dir>          (= tools/pmc_clock_table.py <dir> 5000: the settled long launches)"""
import runpy
import sys

sys.argv = [sys.argv[0], sys.argv[1], "5000"]
runpy.run_path(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "pmc_clock_table.py"), run_name="__main__")
