"""Lesson schedule (reference: utils/curriculum.py:3-59).

`lesson_desc` maps a start iteration (string key) to a list of lessons; a lesson is a list of step names, an int
inside a lesson repeats it. `getLesson(i)` returns lessons[i % n] of the schedule in force at iteration i.
"""


class Curriculum:
    _GAN_ONLY = ("split-style",)

    def __init__(self, lesson_desc):
        self.lessons = []
        valid, evalset = set(), set()
        self.need_sep_gen_opt = False
        self.need_sep_style_ex_opt = False
        self.need_style_in_disc = False
        self.sample_disc = False
        if lesson_desc != 0:
            for start, lessons in lesson_desc.items():
                expanded = []
                for lesson in lessons:
                    repeat = 1
                    steps = []
                    for item in lesson:
                        if isinstance(item, str):
                            self.need_sep_gen_opt |= "auto-style" in item
                            self.need_sep_style_ex_opt |= "style-ex-only" in item
                            self.need_style_in_disc |= "style-super" in item
                            self.sample_disc |= "sample-disc" in item
                            steps.append(item)
                            adversarial = "disc" in item or item in self._GAN_ONLY or "triplet" in item
                            if "gen" not in item and not adversarial:
                                valid.add(item)
                            if not adversarial:
                                evalset.add(item)
                        elif isinstance(item, int):
                            repeat = item
                        else:
                            raise ValueError("unknown thing in lessons: {}".format(item))
                    expanded += [steps] * repeat
                self.lessons.append((int(start), expanded))
        self.lessons.sort(key=lambda t: t[0], reverse=True)   # pop() yields the earliest schedule first
        self.valid = list(valid) + ["valid"]
        self.eval = list(evalset) + ["eval"]

    def getLesson(self, iteration):
        while self.lessons and iteration >= self.lessons[-1][0]:
            self.current_lessons = self.lessons.pop()[1]
        return self.current_lessons[iteration % len(self.current_lessons)]

    def getValid(self):
        return self.valid

    def getEval(self):
        return self.eval
