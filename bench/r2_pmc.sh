#!/bin/bash
bash bench/pmc_step.sh && python tools/pmc_table.py gpurun_out/pmc_step > gpurun_out/r2_pmc_table.md; head -30 gpurun_out/r2_pmc_table.md
