#!/bin/bash
bash tests/r03_sm.sh 2>&1 | grep -A2 "== cfg" | grep -v "^\"Name\|^--"
bash tests/r03_tok3.sh
