#!/usr/bin/env python3
"""Copies the ARFF DATA files the reference's own tests hold (tests/data/arff/**) into tests/golden/arff/, instantiates the two
label templates the way the reference's test utility does (tests/utility.hpp:179-220: LABEL_x_PLACEHOLDER -> -1 / 1 for signed
integers, -1.5 / 1.5 for floating point labels, cat / dog for strings) and writes expected.json with what the reference's tests expect
(tests/detail/io/arff_parsing.cpp: parsed data and labels of the valid files, the error text of every invalid one).  Run in the
build container only: the GPU box has no /root/reference, it uses the committed copies.
usage: python tests/golden/make_arff_fixture.py"""
import glob
import json
import os
import shutil

REF = "/root/reference/tests/data/arff"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "arff")

DENSE = [[-1.117827500607882, -2.9087188881250993, 0.66638344270039144, 1.0978832703949288],
         [-0.5282118298909262, -0.335880984968183973, 0.51687296029754564, 0.54604461446026],
         [0.57650218263054642, 1.01405596624706053, 0.13009428079760464, 0.7261913886869387],
         [-0.20981208921241892, 0.60276937379453293, -0.13086851759108944, 0.10805254527169827],
         [1.88494043717792, 1.00518564317278263, 0.298499933047586044, 1.6464627048813514]]
SPARSE = [[0.0, 0.0, 0.0, 0.0], [0.0, 0.51687296029754564, 0.0, 0.0], [1.01405596624706053, 0.0, 0.0, 0.0],
          [0.60276937379453293, 0.0, -0.13086851759108944, 0.0], [0.0, 0.0, 0.0, 0.298499933047586044]]
LABELS = {"int": ["-1", "1"], "float": ["-1.5", "1.5"], "str": ["cat", "dog"]}

ERRORS = {  # arff_parsing.cpp:62-175 (header), :306-439 (data); "any" = for every label type
    "class_unquoted_nominal_attribute.arff": 'The "@ATTRIBUTE class    0,1" nominal attribute must be enclosed with {}!',
    "class_with_wrong_label.arff": 'May not use the combination of the reserved name "class" and attribute type NUMERIC!',
    "class_without_label.arff": 'The "@ATTRIBUTE class" field must contain class labels!',
    "multiple_classes.arff": "A nominal attribute with the name CLASS may only be provided once!",
    "no_features.arff": "Can't parse file: no feature ATTRIBUTES are defined!",
    "no_data_attribute.arff": "Can't parse file: @DATA is missing!",
    "nominal_attribute_with_wrong_name.arff": 'Read an invalid header entry: "@ATTRIBUTE foo    {0,1}"!',
    "numeric_unquoted.arff": 'A "@ATTRIBUTE second entry   numeric" name that contains a whitespace must be quoted!',
    "numeric_without_name.arff": 'The "@ATTRIBUTE   numeric" field must contain a name!',
    "relation_not_at_beginning.arff": "The @RELATION attribute must be set before any other @ATTRIBUTE!",
    "relation_unquoted.arff": 'A "@RELATION  name with whitespaces" name that contains a whitespace must be quoted!',
    "relation_without_name.arff": 'The "@RELATION" field must contain a name!',
    "wrong_line.arff": 'Read an invalid header entry: "@THIS IS NOT A CORRECT LINE!"!',
    "@_inside_data_section.arff": 'Read @ inside data section!: "@ATTRIBUTE invalid numeric"!',
    "sparse_missing_closing_brace.arff": "Missing closing '}' for sparse data point \"{2 0.51687296029754564,3 0.54604461446026,4 1\" description!",
    "sparse_missing_opening_brace.arff": "Missing opening '{' for sparse data point \"1 0.60276937379453293,2 -0.13086851759108944,4 0}\" description!",
    "sparse_invalid_feature_index.arff": "Trying to add feature/label at index 5 but the maximum index is 4!",
    "sparse_missing_label.arff": 'Missing label for data point "{0 1.88494043717792,1 1.00518564317278263,2 0.298499933047586044,3 1.6464627048813514}"!',
    "dense_missing_value.arff": "Invalid number of features and labels! Found 3 but should be 5!",
    "dense_too_many_values.arff": "Invalid number of features and labels! Found 6 but should be 5!",
    "class_same_label_multiple_times.arff": "Provided 2 labels but only 1 of them was/where unique!",
    "class_with_only_one_label.arff": "Only a single label has been provided!",
}
ERRORS_BY_LABEL_TYPE = {
    "usage_of_undefined_label.arff": {"int": 'Found the label "2" which was not specified in the header ({0,1})!', "str": 'Found the label "2" which was not specified in the header ({0,1})!'},
    "string_label_with_whitespace.arff": {"str": 'String labels may not contain whitespaces, but "Hello World" has at least one!'},
}

if __name__ == "__main__":
    os.makedirs(os.path.join(OUT, "invalid"), exist_ok=True)
    table = {"valid": {}, "invalid": ERRORS, "invalid_by_label_type": ERRORS_BY_LABEL_TYPE}
    for f in glob.glob(os.path.join(REF, "invalid", "*.arff")):
        shutil.copyfile(f, os.path.join(OUT, "invalid", os.path.basename(f)))
    shutil.copyfile(os.path.join(REF, "..", "empty.txt"), os.path.join(OUT, "empty.txt"))
    shutil.copyfile(os.path.join(REF, "3x2_without_label.arff"), os.path.join(OUT, "3x2_without_label.arff"))
    table["valid"]["3x2_without_label.arff"] = {"data": [[1.5, -2.9], [0.0, -0.3], [5.5, 0.0]], "labels": None, "label_type": "int"}
    for template, data in (("5x4_TEMPLATE.arff", DENSE), ("5x4_sparse_TEMPLATE.arff", SPARSE)):
        text = open(os.path.join(REF, template)).read()
        for kind, (a, b) in LABELS.items():
            name = template.replace("TEMPLATE", kind)
            with open(os.path.join(OUT, name), "w") as f:
                f.write(text.replace("LABEL_1_PLACEHOLDER", a).replace("LABEL_2_PLACEHOLDER", b))
            table["valid"][name] = {"data": data, "labels": [a, a, b, b, b], "label_type": kind}
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print(f"wrote fixtures to {OUT}")
