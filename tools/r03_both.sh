#!/bin/bash
bash tools/r03_sm.sh 2>&1 | grep -A2 "== cfg" | grep -v "^\"Name\|^--"
bash tools/r03_tok3.sh
